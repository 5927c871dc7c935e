#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's own numpy/scipy functions.

Runs only in the build container (needs /root/reference).  The reference modules
`import cv2` at the top although the functions exercised here never call it; cv2
is not installed, so an EMPTY module object named cv2 is registered first -- it
provides no functionality.  Only inputs and outputs are stored (fixtures are
data); no reference source is copied.

Functions exercised (reference file:line):
  BundleAdjustment._rotate / _reproject / _fun      vo_pipeline/bundleAdjustment.py:14-45
  BundleAdjustment._bundle_adjustment_sparsity      vo_pipeline/bundleAdjustment.py:47-67
  BundleAdjustment._preprocess_data / bundle_adjustment   vo_pipeline/bundleAdjustment.py:69-134  (these two call
      cv.Rodrigues; for them, and only for them, the stand-in module gets a Rodrigues built on
      scipy.spatial.transform.Rotation -- the conversion, not OpenCV's code, so the fixtures record it in `rodrigues`)
  scipy 2-point FD Jacobian through that sparsity   (what least_squares does at :96-97)
  BootstrapInitializer._linear_triangulation        vo_pipeline/bootstrap.py:141-156
  BootstrapInitializer._transform_matrix            vo_pipeline/bootstrap.py:76-138
  ContinuousVO._baseline_uncertainty                vo_pipeline/continuousVO.py:249-271
  utils.matrix.hom_inv / skew                       utils/matrix.py:4-40
  KeypointTrajectories.create_trajectory / tracked_to / at_frame   vo_pipeline/keypointTrajectory.py:42-138
  FrameQueue.add / get, ContinuousVO._bundle_adjustment            utils/frameQueue.py:20-30, continuousVO.py:273-344
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def rodrigues(rv):
    th = np.linalg.norm(rv)
    if th == 0:
        return np.eye(3)
    k = rv / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx


def ba_problem(F, P, vis, seed, K):
    """Seeded scene: points in front of cameras moving along z; observations with 0.5 px noise."""
    rng = np.random.default_rng(seed)
    pts = rng.uniform([-20, -3, 6], [20, 3, 60], (P, 3))
    cams = np.zeros((F, 6))
    cams[:, :3] = rng.normal(0, 0.01, (F, 3))
    cams[:, 5] = -0.8 * np.arange(F)
    cams[:, 3:5] = rng.normal(0, 0.05, (F, 2))
    ci, pi, obs = [], [], []
    for f in range(F):
        R = rodrigues(cams[f, :3])
        for p in range(P):
            if rng.uniform() > vis:
                continue
            Xc = R @ pts[p] + cams[f, 3:]
            uv = (K @ Xc)[:2] / Xc[2]
            ci.append(f); pi.append(p); obs.append(uv + rng.normal(0, 0.5, 2))
    ci = np.asarray(ci, np.int64); pi = np.asarray(pi, np.int64)
    obs = np.asarray(obs, np.float32)
    # perturbed parameter vector (what the optimiser would be evaluated at)
    x = np.hstack([(cams + rng.normal(0, 1e-3, cams.shape)).ravel(), (pts + rng.normal(0, 0.05, pts.shape)).ravel()])
    return x, ci, pi, obs


def main():
    os.makedirs(OUT, exist_ok=True)
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, REF)
    sys.modules["cv2"] = types.ModuleType("cv2")   # empty: nothing below calls into it
    from vo_pipeline.bundleAdjustment import BundleAdjustment
    from vo_pipeline.bootstrap import BootstrapInitializer
    from vo_pipeline.continuousVO import ContinuousVO
    from utils.matrix import hom_inv, skew
    from scipy.optimize._numdiff import approx_derivative, group_columns

    K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1]], dtype=np.float32)
    ba = BundleAdjustment(K)

    # ---- BA residual / sparsity / FD Jacobian --------------------------------
    for name, (F, P, vis, seed) in {"ba_small": (3, 20, 1.0, 0), "ba_mid": (6, 150, 0.7, 1),
                                    "ba_frozen8": (8, 60, 0.8, 2)}.items():
        x, ci, pi, obs = ba_problem(F, P, vis, seed, K.astype(np.float64))
        r = ba._fun(x, F, P, ci, pi, obs)
        A = ba._bundle_adjustment_sparsity(F, P, ci, pi)
        groups = group_columns(A)
        J = approx_derivative(ba._fun, x, method="2-point", sparsity=(A, groups), args=(F, P, ci, pi, obs)).tocsr()
        Acsr = A.tocsr()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), F=F, P=P, K=K, x=x, ci=ci, pi=pi, obs=obs, r=r,
                            cost=0.5 * float(r @ r), A_indices=Acsr.indices, A_indptr=Acsr.indptr,
                            J_data=J.data, J_indices=J.indices, J_indptr=J.indptr,
                            n_fixed=int(F / 4) + 1)
        print(name, "O =", ci.size, "cost =", 0.5 * float(r @ r), "nnz(A) =", Acsr.nnz)

    # ---- outlier pre-filter and the whole solve, on two seeded windows with planted outliers -----
    # Inputs as ContinuousVO._bundle_adjustment hands them over (continuousVO.py:309-313): float32 poses, landmarks and
    # keypoints, int64 indices sorted by camera.  Stored: what _preprocess_data keeps, what bundle_adjustment returns,
    # and the cost 0.5*|f|^2 of the reference's own residual at its start and at its solution (kept observations).
    from scipy.spatial.transform import Rotation

    def rodrigues_shim(x):
        x = np.asarray(x, np.float64)
        if x.size == 3:
            return Rotation.from_rotvec(x.ravel()).as_matrix(), None
        return Rotation.from_matrix(x).as_rotvec().reshape(3, 1), None

    sys.modules["cv2"].Rodrigues = rodrigues_shim
    for name, (F, P, vis, seed, n_out) in {"ba_window_a": (6, 200, 0.8, 11, 25), "ba_window_b": (10, 400, 0.6, 12, 60)}.items():
        x, ci, pi, obs = ba_problem(F, P, vis, seed, K.astype(np.float64))
        rng = np.random.default_rng(seed + 100)
        bad = rng.choice(ci.size, n_out, replace=False)
        obs = obs.copy()
        obs[bad] += (rng.uniform(12, 80, (n_out, 2)) * rng.choice([-1, 1], (n_out, 2))).astype(np.float32)
        cams = x[:6 * F].reshape(F, 6)
        M = np.zeros((F, 3, 4), np.float32)
        for f in range(F):
            M[f, :, :3] = rodrigues(cams[f, :3])
            M[f, :, 3] = cams[f, 3:]
        landmarks = x[6 * F:].reshape(P, 3).astype(np.float32)
        kp_kept, ci_kept, pi_kept = ba._preprocess_data(M, landmarks, ci, pi, obs)
        adj_M, adj_L = ba.bundle_adjustment(M, landmarks, ci, pi, obs)

        def cost_at(Mx, Lx):
            cp = np.zeros((F, 6))
            for f in range(F):
                cp[f, :3] = rodrigues_shim(Mx[f, :, :3])[0].ravel()
                cp[f, 3:] = Mx[f, :, 3]
            r = ba._fun(np.hstack([cp.ravel(), np.asarray(Lx, np.float64).ravel()]), F, P, ci_kept, pi_kept, kp_kept)
            return 0.5 * float(r @ r)

        c0, c1 = cost_at(M, landmarks), cost_at(adj_M, adj_L)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), F=F, P=P, K=K, M=M, landmarks=landmarks, ci=ci, pi=pi, keypoints=obs,
                            planted=np.sort(bad), kept_keypoints=kp_kept, kept_ci=ci_kept, kept_pi=pi_kept, adjusted_M=adj_M,
                            adjusted_landmarks=adj_L, cost_start=c0, cost_final=c1, n_fixed=int(F / 4) + 1, threshold=100.0,
                            rodrigues="scipy.spatial.transform.Rotation stand-in for cv.Rodrigues")
        print(name, "O =", ci.size, "kept =", ci_kept.size, "cost", c0, "->", c1,
              "frozen cameras moved by", float(np.abs(adj_M[:int(F / 4) + 1] - M[:int(F / 4) + 1]).max()))
    del sys.modules["cv2"].Rodrigues

    # ---- _rotate edge cases (theta == 0 -> points unchanged) ------------------
    rng = np.random.default_rng(3)
    rv = rng.normal(0, 1, (12, 3)); rv[0] = 0; rv[1] = [1e-12, 0, 0]; rv[2] = [0, np.pi, 0]; rv[3] = [1e-7, -1e-7, 2e-7]
    X = rng.normal(0, 10, (12, 3))
    np.savez_compressed(os.path.join(OUT, "rotate.npz"), rv=rv, X=X, out=ba._rotate(X, rv))

    # ---- two-view geometry ----------------------------------------------------
    rng = np.random.default_rng(4)
    Kd = K.astype(np.float64)
    n = 40
    Pw = rng.uniform([-8, -3, 8], [8, 3, 40], (n, 3))
    R = rodrigues(np.array([0.01, -0.03, 0.005])); t = np.array([0.9, 0.05, 0.2]); t /= np.linalg.norm(t)
    M1 = (K @ np.eye(3, 4, dtype=np.float32))
    M2 = (Kd @ np.hstack([R, t[:, None]]))
    p1 = (Kd @ Pw.T).T; p1 /= p1[:, 2:]
    p2 = (Kd @ (R @ Pw.T + t[:, None])).T; p2 /= p2[:, 2:]
    p1 = p1.astype(np.float32); p2 = p2.astype(np.float32)
    tri = BootstrapInitializer._linear_triangulation(p1, p2, M1, M2)
    tx = skew(t.astype(np.float32)).astype(np.float64)
    E = tx @ R
    Fm = np.linalg.inv(Kd).T @ E @ np.linalg.inv(Kd)
    boot = object.__new__(BootstrapInitializer)
    boot.K = K; boot.max_point_dist = 50
    T, cloud, mask = boot._transform_matrix(Fm, p1, p2)
    np.savez_compressed(os.path.join(OUT, "two_view.npz"), K=K, p1=p1, p2=p2, M1=M1, M2=M2, tri=tri, F=Fm, T=T,
                        cloud=cloud, mask=mask, R_true=R, t_true=t, max_point_dist=50)
    print("two_view: tri err", np.abs(tri[:, :3] - Pw).max(), "mask", mask.sum())

    # ---- baseline uncertainty / hom_inv --------------------------------------
    vo = object.__new__(ContinuousVO)
    T0 = np.eye(4); T0[:3, :3] = rodrigues(np.array([0.02, 0.01, -0.03])); T0[:3, 3] = [0.1, -0.2, 0.3]
    T1 = np.eye(4); T1[:3, :3] = rodrigues(np.array([0.03, -0.02, 0.01])); T1[:3, 3] = [0.5, -0.1, -1.2]
    lm = rng.uniform([-10, -3, -5], [10, 3, 40], (30, 3)).astype(np.float32)
    bu = vo._baseline_uncertainty(T0, T1, lm)
    bu_none = vo._baseline_uncertainty(T0, T1, lm * np.array([1, 1, -1], np.float32) - np.array([0, 0, 100], np.float32))
    np.savez_compressed(os.path.join(OUT, "baseline.npz"), T0=T0, T1=T1, landmarks=lm, uncertainty=bu,
                        uncertainty_none=bu_none, T0_inv=hom_inv(T0), skew_v=np.array([1.0, 2.0, 3.0]),
                        skew_m=skew(np.array([1.0, 2.0, 3.0])))
    print("baseline", bu, bu_none)

    # ---- track store and BA window marshalling --------------------------------
    # A seeded script of store operations is replayed on the reference's KeypointTrajectories;
    # ops rows: (kind 0=create 1=tracked_to, frame, trajectory id or -1, x, y, landmark id or -1).
    from vo_pipeline.keypointTrajectory import KeypointTrajectories
    from vo_pipeline.frameState import FrameState
    from utils.frameQueue import FrameQueue
    rng = np.random.default_rng(9)
    store = KeypointTrajectories()
    ops, lms = [], []
    alive = []
    out = {}

    def create(frame, with_landmark):
        pt = rng.uniform(0, 600, 2).astype(np.float32)
        lid = -1
        if with_landmark:
            lid = len(store.landmarks)
            lm = rng.uniform(-5, 30, 3).astype(np.float32)
            store.landmarks.append(lm); lms.append(lm)
        _, t = store.create_trajectory(frame, pt, np.eye(4), None if lid < 0 else lid)
        ops.append((0, frame, -1, pt[0], pt[1], lid)); alive.append(t)

    for _ in range(12):
        create(3, True)
    create(3, False)
    for frame in range(4, 10):
        rng.shuffle(alive)
        alive = [t for t in alive if rng.uniform() > 0.12]
        for t in alive:
            pt = rng.uniform(0, 600, 2).astype(np.float32)
            store.tracked_to(t, frame, pt, np.eye(4))
            ops.append((1, frame, t, pt[0], pt[1], -1))
        if frame in (6, 8):
            for _ in range(5):
                create(frame, True)
    for frame in range(3, 10):
        kp, tr, lm = store.at_frame(frame)
        out[f"f{frame}_kp"], out[f"f{frame}_traj"], out[f"f{frame}_lm"] = kp, tr, lm
    out["ops"] = np.asarray(ops, np.float64)
    out["landmarks"] = np.asarray(lms, np.float32)

    class Capture:
        def bundle_adjustment(self, M, landmarks, ci, pi, kp):
            self.args = (M, landmarks, ci, pi, kp)
            adj = M.astype(np.float64).copy(); adj[:, :, 3] += 0.25
            return adj, landmarks.astype(np.float64) + 0.5

    vo = object.__new__(ContinuousVO)
    vo.keypoint_trajectories = store
    vo.frame_queue = FrameQueue(50)
    poses = []
    for f in range(0, 9):
        Tf = np.eye(4); Tf[:3, :3] = rodrigues(rng.normal(0, 0.02, 3)); Tf[:3, 3] = rng.normal(0, 1, 3)
        poses.append(Tf)
        vo.frame_queue.add(FrameState(f, np.zeros((2, 2), np.uint8), Tf.copy()))
    vo.keyframes = [vo.frame_queue.queue[4], vo.frame_queue.queue[7]]     # keyframes at frames 4 and 7
    vo.bundle_adjustment = Capture()
    Tcur = np.eye(4); Tcur[:3, 3] = [0.3, -0.1, 2.0]
    ret = vo._bundle_adjustment(9, Tcur)
    M, L, ci, pi, kp = vo.bundle_adjustment.args
    out.update(win_queue_poses=np.asarray(poses), win_T=Tcur, win_M=M, win_landmarks=L, win_ci=ci, win_pi=pi,
               win_kp=kp, win_ret=ret, win_poses_after=np.asarray([fs.pose for fs in vo.frame_queue.queue]),
               win_landmarks_after=np.asarray(store.landmarks, np.float64), win_keyframes=np.array([4, 7]))
    np.savez_compressed(os.path.join(OUT, "track_store.npz"), **out)
    print("track_store: ops", len(ops), "window O =", ci.size, "F =", M.shape[0], "P =", L.shape[0])


if __name__ == "__main__":
    main()
