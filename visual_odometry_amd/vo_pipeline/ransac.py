"""RANSAC drivers whose candidate scoring runs on the GPU.

Stand-ins for the two cv2 calls of the hot path:
  cv2.findFundamentalMat(pts0, pts1, cv2.FM_RANSAC, thr, conf, maxIters)   vo_pipeline/bootstrap.py:246-252
  cv.solvePnPRansac(X, x, K, None, flags=SOLVEPNP_P3P, ...)                vo_pipeline/poseEstimation.py:61-69

Structure: the host draws every minimal sample up front with OpenCV's generator
and subset rule,
the GPU solves all minimal problems (vo_seven_point_solve / vo_p3p_solve, one
sample per lane) and scores every candidate against every correspondence in one
launch (vo_score_fundamental / vo_score_pnp), and the host replays OpenCV's
sequential loop over the counts -- best-so-far update, adaptive iteration count,
early stop -- so the outcome equals that loop run on the same candidate sequence.
find_fundamental_mat / solve_pnp_ransac make one library call each
(vo_find_fundamental_ransac / vo_solve_pnp_ransac: candidates stay on the device,
the loop is replayed in C); the *_stepwise functions are the same computation
spelled out over the single-step entry points, for callers that bring their own
candidates and as the executable statement the parity tests hold the fused calls to.
The minimal solvers are independent implementations (same geometry, different
numerics than OpenCV's), so candidates agree with cv2 only to solver precision.
"""
from typing import Optional, Tuple

import numpy as np

from .. import hip
from . import _runtime


class CvRNG:
    """cv::RNG: multiply-with-carry, state = (uint32)state * 4164903690 + (state >> 32)."""
    COEFF = 4164903690

    def __init__(self, state: int = 0xFFFFFFFFFFFFFFFF):
        self.state = state & 0xFFFFFFFFFFFFFFFF

    def next(self) -> int:
        self.state = ((self.state & 0xFFFFFFFF) * self.COEFF + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return self.state & 0xFFFFFFFF

    def uniform(self, a: int, b: int) -> int:
        return a if a == b else a + self.next() % (b - a)


def ransac_update_num_iters(p: float, ep: float, model_points: int, max_iters: int) -> int:
    """RANSACUpdateNumIters: iterations needed for confidence p at outlier ratio ep."""
    p = min(max(p, 0.0), 1.0)
    ep = min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, np.finfo(np.float64).tiny)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < np.finfo(np.float64).tiny:
        return 0
    num, denom = np.log(num), np.log(denom)
    return max_iters if (denom >= 0 or -num >= max_iters * (-denom)) else int(round(num / denom))


def draw_subsets(rng: CvRNG, count: int, model_points: int, n_subsets: int, m1: Optional[np.ndarray] = None,
                 m2: Optional[np.ndarray] = None, max_attempts: int = 10000) -> np.ndarray:
    """n_subsets index sets of distinct indices drawn as RANSACPointSetRegistrator::getSubset does;
    with m1/m2 given, subsets whose last point is collinear with an earlier pair are redrawn
    (host-side routine of the library; advances rng)."""
    out, rng.state = hip.ransac_draw_subsets(rng.state, count, model_points, n_subsets, m1, m2, max_attempts)
    return out


def replay_ransac_loop(counts: np.ndarray, owner: np.ndarray, n: int, model_points: int, conf: float,
                       max_iters: int) -> int:
    """Replay OpenCV's loop over pre-scored candidates.  counts[c] = inliers of candidate c,
    owner[c] = iteration that produced it (ascending).  Returns the winning candidate or -1."""
    niters, best, best_c = max_iters, -1, -1
    c, C = 0, counts.shape[0]
    it = 0
    while it < niters:
        while c < C and owner[c] < it:
            c += 1
        while c < C and owner[c] == it:
            good = int(counts[c])
            if good > max(best, model_points - 1):
                best, best_c = good, c
                niters = ransac_update_num_iters(conf, (n - good) / n, model_points, niters)
            c += 1
        it += 1
    return best_c


# ---------------------------------------------------------------- F by RANSAC
def find_fundamental_mat(pts0: np.ndarray, pts1: np.ndarray, thr: float, conf: float, max_iters: int,
                         device: int = 0) -> Tuple[Optional[np.ndarray], np.ndarray]:
    """-> (F (3,3) float64 or None, mask (n,1) uint8), as cv2.findFundamentalMat(..., FM_RANSAC, ...)."""
    F, mask, _, _ = hip.find_fundamental_ransac(_runtime.context(device), pts0, pts1, thr, conf, int(max_iters))
    return F, mask.reshape(-1, 1)


def find_fundamental_mat_stepwise(pts0: np.ndarray, pts1: np.ndarray, thr: float, conf: float, max_iters: int,
                                  device: int = 0) -> Tuple[Optional[np.ndarray], np.ndarray]:
    """find_fundamental_mat out of the single-step entry points (sampler, 7-point solver, scoring, replay)."""
    m1 = np.ascontiguousarray(pts0, np.float32).reshape(-1, 2)
    m2 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2)
    n = m1.shape[0]
    if m2.shape[0] != n:
        raise ValueError("find_fundamental_mat: point sets differ in length")
    if n < 8:  # OpenCV needs >= 7 and switches estimators below 15; keep the RANSAC form for n >= 8
        return None, np.zeros((n, 1), np.uint8)
    rng = CvRNG()
    subsets = draw_subsets(rng, n, 7, int(max_iters), m1, m2)
    ctx = _runtime.context(device)
    Fall, nsol = hip.seven_point_solve(ctx, m1, m2, subsets)          # (S,3,3,3), (S,)
    slot = np.arange(3)[None, :] < nsol[:, None]                      # real solutions per sample
    owner = np.repeat(np.arange(subsets.shape[0]), 3)[slot.ravel()]
    Fs = Fall.reshape(-1, 3, 3)[slot.ravel()]
    if Fs.shape[0] == 0:
        return None, np.zeros((n, 1), np.uint8)
    counts, _ = hip.score_fundamental(ctx, m1, m2, Fs, thr, want_masks=False)
    win = replay_ransac_loop(counts, owner, n, 7, conf, int(max_iters))
    if win < 0:
        return None, np.zeros((n, 1), np.uint8)
    _, mask = hip.score_fundamental(ctx, m1, m2, Fs[win:win + 1], thr, want_masks=True)
    return Fs[win].copy(), mask.reshape(n, 1)


# ---------------------------------------------------------------- PnP by RANSAC
def refine_pose(ctx, X: np.ndarray, x: np.ndarray, K: np.ndarray, R: np.ndarray, t: np.ndarray, iters: int = 10,
                mask: Optional[np.ndarray] = None):
    """Gauss-Newton on the reprojection error over the correspondences selected by mask (pose only), one
    GPU launch (vo_refine_pose) -> (R (3,3), t (3,))."""
    from scipy.spatial.transform import Rotation
    rvec, t2, _ = hip.refine_pose(ctx, X, x, K, Rotation.from_matrix(R).as_rotvec(), t, mask, iters)
    return Rotation.from_rotvec(rvec).as_matrix(), t2


def solve_pnp_ransac(X: np.ndarray, x: np.ndarray, K: np.ndarray, thr: float, conf: float, max_iters: int,
                     device: int = 0):
    """-> (success, R (3,3), t (3,), inliers (k,1) int32), as cv.solvePnPRansac(flags=SOLVEPNP_P3P)
    followed by its refit on the inliers (OpenCV refits with EPnP; here Gauss-Newton on the same error)."""
    Rt, mask, good, _ = hip.solve_pnp_ransac(_runtime.context(device), X, x, K, thr, conf, int(max_iters), refine_iters=10)
    if Rt is None:
        return False, np.eye(3), np.zeros(3), np.zeros((0, 1), np.int32)
    return True, Rt[:, :3].copy(), Rt[:, 3].copy(), np.nonzero(mask)[0].astype(np.int32).reshape(-1, 1)


def solve_pnp_ransac_stepwise(X: np.ndarray, x: np.ndarray, K: np.ndarray, thr: float, conf: float, max_iters: int,
                              device: int = 0):
    """solve_pnp_ransac out of the single-step entry points (sampler, P3P solver, scoring, replay, refit)."""
    X = np.ascontiguousarray(X, np.float32).reshape(-1, 3)
    x = np.ascontiguousarray(x, np.float32).reshape(-1, 2)
    n = X.shape[0]
    if x.shape[0] != n:
        raise ValueError("solve_pnp_ransac: point sets differ in length")
    empty = np.zeros((0, 1), np.int32)
    if n < 4:
        return False, np.eye(3), np.zeros(3), empty
    Kd = np.asarray(K, np.float64).reshape(3, 3)
    rng = CvRNG()
    subsets = draw_subsets(rng, n, 4, int(max_iters))
    ctx = _runtime.context(device)
    Rt_all, valid = hip.p3p_solve(ctx, X, x, Kd, subsets)              # one candidate per sample
    owner = np.nonzero(valid)[0].astype(np.int64)
    if owner.size == 0:
        return False, np.eye(3), np.zeros(3), empty
    Rt = Rt_all[owner]
    counts, _ = hip.score_pnp(ctx, X, x, Kd, Rt, thr, want_masks=False)
    win = replay_ransac_loop(counts, owner, n, 4, conf, int(max_iters))
    if win < 0:
        return False, np.eye(3), np.zeros(3), empty
    _, mask = hip.score_pnp(ctx, X, x, Kd, Rt[win:win + 1], thr, want_masks=True)
    inl = np.nonzero(mask[0])[0]
    R, t = Rt[win][:, :3], Rt[win][:, 3]
    if inl.size >= 4:
        R2, t2 = refine_pose(ctx, X, x, Kd, R, t, mask=mask[0])
        if np.all(np.isfinite(R2)) and np.all(np.isfinite(t2)):
            R, t = R2, t2
    return True, R, t, inl.astype(np.int32).reshape(-1, 1)
