"""BundleAdjustment -- API of the reference's vo_pipeline/bundleAdjustment.py:9-134.

_fun / _reproject / _rotate evaluate on the GPU (vo_ba_residual).  The solve keeps
the reference's problem definition -- outlier pre-filter, parameter vector
[F x (rvec, t) | P x XYZ], first int(F/4)+1 cameras and their observation rows
frozen, cost 0.5*|r|^2, ftol 1e-4 -- but where the reference hands scipy's TRF a
finite-difference Jacobian, this runs Levenberg-Marquardt on the analytic normal
equations with a Schur complement on the landmarks, all on the GPU, the loop's
accept / reject decisions included (vo_ba_solve: trial steps are queued several at a
time and the host reads the loop's record once per round).  Iterates therefore differ
from scipy's; the residual, cost and Jacobian at a given x are the same.  A window ends
when an accepted step gains less than ftol of the cost, when a rejected step moved it by
less than that (`stop_on_flat_reject`, on by default: with the frozen cameras'
observations in the cost but not in the Jacobian a window typically ends in a run of
rejected steps whose cost change shrinks tenfold per try; scipy has no such rule),
or after ten rejected steps in a row.
"""
from typing import Tuple

import numpy as np
from scipy.sparse import coo_matrix
from scipy.spatial.transform import Rotation

from .. import hip, params
from . import _runtime


class _AdjustmentInFlight:
    """A window between BundleAdjustment.bundle_adjustment_begin and its result."""

    def __init__(self, owner, pose_shape, landmark_shape, x, solver):
        self.owner, self.pose_shape, self.landmark_shape, self.x, self.solver = owner, pose_shape, landmark_shape, x, solver

    def finish(self) -> Tuple[np.ndarray, np.ndarray]:
        """(adjusted_M (F,3,4) float64, adjusted_landmarks (P,3) float64); waits for the window's trial steps."""
        x, solver, self.solver = self.x, self.solver, None
        if solver is not None:
            try:
                x, self.owner.last_stats = solver.solve_finish(self.owner._workspace)     # copied out below
            finally:
                solver.close()
        F = self.pose_shape[0]
        cams = x[:6 * F].reshape(F, 6)
        adjusted_M = np.zeros(self.pose_shape)
        adjusted_M[:, :, 3] = cams[:, 3:]
        if F:
            adjusted_M[:, :, :3] = Rotation.from_rotvec(cams[:, :3]).as_matrix()
        return adjusted_M, x[6 * F:].reshape(self.landmark_shape).copy()


class BundleAdjustment:

    def __init__(self, K: np.ndarray):
        self.K = K

    @property
    def _workspace(self) -> "hip.HostWorkspace":
        """Host buffers of a window's parameter vectors, in and out: reused from window to window, not re-faulted."""
        return self.__dict__.setdefault("_host_workspace", hip.HostWorkspace())

    # ---- residual pieces, same signatures as the reference ------------------
    def _rotate(self, points, rot_vecs):
        """Rodrigues rotation of points (N,3) by rotation vectors (N,3); a zero vector leaves the
        point unchanged.  Host-side helper (the batched residual path is _fun / _reproject)."""
        points = np.asarray(points, np.float64)
        rot_vecs = np.asarray(rot_vecs, np.float64)
        theta = np.sqrt((rot_vecs ** 2).sum(axis=1, keepdims=True))
        axis = np.divide(rot_vecs, theta, out=np.zeros_like(rot_vecs), where=theta != 0)
        along = (points * axis).sum(axis=1, keepdims=True)
        c, s = np.cos(theta), np.sin(theta)
        return c * points + s * np.cross(axis, points) + along * (1 - c) * axis

    def _reproject(self, points, camera_params):
        """Project 3-D points (N,3) with per-point camera parameters (N,6) -> (N,2) pixels."""
        ctx = _runtime.context()
        points = np.asarray(points, np.float64)
        camera_params = np.asarray(camera_params, np.float64)
        n = points.shape[0]
        idx = np.arange(n, dtype=np.int64)
        x = np.hstack([camera_params.ravel(), points.ravel()])
        r, _ = hip.ba_residual(ctx, x, n, n, idx, idx, np.zeros((n, 2)), self.K)
        return r.reshape(n, 2)

    def _fun(self, params, n_frames, n_points, camera_indices, point_indices, points_2d):
        """Residuals (2*O,) float64, interleaved (u0, v0, u1, v1, ...)."""
        ctx = _runtime.context()
        r, _ = hip.ba_residual(ctx, params, int(n_frames), int(n_points), camera_indices, point_indices,
                               np.asarray(points_2d, np.float64), self.K)
        return r

    def _bundle_adjustment_sparsity(self, n_cameras, n_points, camera_indices, point_indices):
        """Jacobian sparsity (2*O x 6F+3P) with the rows of the first int(F/4)+1 cameras' observations empty."""
        camera_indices = np.asarray(camera_indices)
        point_indices = np.asarray(point_indices)
        m = camera_indices.size * 2
        n = n_cameras * 6 + n_points * 3
        start = self._obs_start(n_cameras, camera_indices)
        obs = np.arange(start, camera_indices.size)
        rows, cols = [], []
        for s in range(6):
            for odd in (0, 1):
                rows.append(2 * obs + odd)
                cols.append(camera_indices[obs] * 6 + s)
        for s in range(3):
            for odd in (0, 1):
                rows.append(2 * obs + odd)
                cols.append(n_cameras * 6 + point_indices[obs] * 3 + s)
        rows = np.concatenate(rows) if rows else np.zeros(0, int)
        cols = np.concatenate(cols) if cols else np.zeros(0, int)
        return coo_matrix((np.ones(rows.size, dtype=int), (rows, cols)), shape=(m, n)).tolil()

    @staticmethod
    def _obs_start(n_cameras, camera_indices) -> int:
        """Number of leading observations whose Jacobian rows stay empty: those counted for the
        first int(F/4)+1 cameras (observations are expected sorted by camera)."""
        ci = np.asarray(camera_indices)
        return int(np.count_nonzero((ci >= 0) & (ci < int(n_cameras / 4) + 1)))

    @staticmethod
    def _camera_params(M: np.ndarray) -> np.ndarray:
        """(F, 6) rotation vector and translation of every pose (bundleAdjustment.py:86-91), in one host-side library call."""
        return hip.ba_camera_params(M)

    def _preprocess_data(self, M, landmarks, camera_indices, point_indices, keypoints):
        """Drop observations whose squared reprojection error at the start exceeds BA_DISTANCE_TH."""
        n_points = landmarks.shape[0]
        num_frames = M.shape[0]
        x0 = np.hstack((self._camera_params(M).ravel(), np.asarray(landmarks, np.float64).ravel()))
        r = self._fun(x0, num_frames, n_points, camera_indices, point_indices, keypoints)
        u, v = r[0::2], r[1::2]
        difference = u * u + v * v
        drop = difference > params.BA_DISTANCE_TH
        if not drop.any():
            return keypoints, camera_indices, point_indices
        kept = np.flatnonzero(~drop)
        return keypoints[kept], camera_indices[kept], point_indices[kept]

    # ---- solve ------------------------------------------------------------------
    def bundle_adjustment(self, M: np.ndarray, landmarks: np.ndarray, camera_indices: np.ndarray,
                          point_indices: np.ndarray, keypoints: np.ndarray,
                          max_iters: int = 50, ftol: float = 1e-4, stop_on_flat_reject: bool = True,
                          device_loop: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        """
        :param M:               (F, 3, 4) camera poses [R|t]
        :param landmarks:       (P, 3) point cloud
        :param camera_indices:  (O,) frame index of each observation, ascending
        :param point_indices:   (O,) landmark index of each observation
        :param keypoints:       (O, 2) observed pixels
        :param stop_on_flat_reject: end the window on a rejected step that moved the cost by less than ftol
                                (False: keep raising the damping, up to ten rejected steps in a row)
        :param device_loop:     the loop's decisions on the GPU (vo_ba_solve); False: the same loop on the host, one
                                cost read-back per trial step
        :return (adjusted_M (F,3,4) float64, adjusted_landmarks (P,3) float64)
        """
        return self.bundle_adjustment_begin(M, landmarks, camera_indices, point_indices, keypoints, max_iters, ftol,
                                            stop_on_flat_reject, device_loop).finish()

    def bundle_adjustment_begin(self, M: np.ndarray, landmarks: np.ndarray, camera_indices: np.ndarray,
                                point_indices: np.ndarray, keypoints: np.ndarray,
                                max_iters: int = 50, ftol: float = 1e-4, stop_on_flat_reject: bool = True,
                                device_loop: bool = True, first_round: int = 0) -> "_AdjustmentInFlight":
        """bundle_adjustment in two halves (not in the reference): this one builds the window's solver and queues its
        first round of trial steps without waiting for them; `.finish()` of what it returns gives bundle_adjustment's
        result.  In between the caller may run other GPU work on another context (_runtime.side_context); with
        such work to hide behind it asks for a longer first round of queued trial steps (first_round, 0: five)."""
        camera_indices = np.asarray(camera_indices, np.int64)
        point_indices = np.asarray(point_indices, np.int64)
        F, P = M.shape[0], landmarks.shape[0]
        x = self._workspace.take("start x", 6 * F + 3 * P, np.float64)
        x[:6 * F] = self._camera_params(M).ravel()
        x[6 * F:] = np.asarray(landmarks).reshape(-1)
        solver = None
        if camera_indices.size and F and P:
            # the outlier pre-filter of _preprocess_data, the compaction of what it keeps and the solver's observation
            # lists in one library call: the observations cross to the device once
            solver = hip.BaSolver.filtered(_runtime.context(), F, P, camera_indices, point_indices,
                                           keypoints, self.K, x, params.BA_DISTANCE_TH, int(F / 4) + 1)
            if not device_loop:
                x = self._levenberg_marquardt(x, solver, max_iters, ftol, stop_on_flat_reject)      # closes the solver
                solver = None
            elif solver.O:
                try:
                    solver.solve_begin(x, max_iters, ftol, 1e-3, stop_on_flat_reject, first_round)
                except Exception:
                    solver.close()
                    raise
            else:
                solver.close()
                solver = None
        # without a solver in flight the start is the result: a copy, the workspace's buffer is the next window's
        return _AdjustmentInFlight(self, M.shape, landmarks.shape, x if solver is not None else x.copy(), solver)

    def _levenberg_marquardt(self, x, solver, max_iters, ftol, stop_on_flat_reject=True):
        """The LM loop with the decisions on the host (what vo_ba_solve runs on the device): every trial step (Schur
        complement, reduced solve, back-substitution, re-linearisation) runs on the GPU, the host compares two costs
        after each.  Closes the solver."""
        try:
            if solver.O == 0:
                return x
            cost = solver.linearize(x)
            lam = 1e-3
            for _ in range(max_iters):
                improved = converged = False
                prev = cost
                for _try in range(10):
                    cost_new, solved = solver.try_step(lam)
                    usable = solved and np.isfinite(cost_new)
                    if usable and cost_new < cost:
                        solver.accept()
                        cost = cost_new
                        lam = max(lam / 3, 1e-9)
                        improved = True
                        break
                    # A rejected step that moves the cost by less than the tolerance: stronger damping only shortens
                    # the step along (nearly) the same direction, so nothing above the tolerance is left to gain.
                    # (The frozen cameras' observations count in the cost but not in the Jacobian -- the reference's
                    # sparsity pattern --, so the model's minimum is not exactly the cost's and such endings are common.)
                    if stop_on_flat_reject and usable and abs(cost_new - cost) < ftol * cost:
                        converged = True
                        break
                    lam *= 10
                if converged or not improved or (prev - cost) < ftol * prev:
                    break
            return solver.params()
        finally:
            solver.close()
