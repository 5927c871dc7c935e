"""RANSAC drivers whose candidate scoring runs on the GPU.

Stand-ins for the two cv2 calls of the hot path:
  cv2.findFundamentalMat(pts0, pts1, cv2.FM_RANSAC, thr, conf, maxIters)   vo_pipeline/bootstrap.py:246-252
  cv.solvePnPRansac(X, x, K, None, flags=SOLVEPNP_P3P, ...)                vo_pipeline/poseEstimation.py:61-69

Structure: the host draws every minimal sample up front with OpenCV's generator
and subset rule, solves all minimal problems in batched numpy (7-point F, P3P),
the GPU scores every candidate against every correspondence in one launch
(vo_score_fundamental / vo_score_pnp), and the host replays OpenCV's sequential
loop over the counts -- best-so-far update, adaptive iteration count, early stop
-- so the outcome equals that loop run on the same candidate sequence.  The
minimal solvers are independent implementations (same geometry, different
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


def _collinear_with_last(pts: np.ndarray) -> bool:
    """OpenCV's haveCollinearPoints: is the last point (nearly) on a line through two earlier ones."""
    i = pts.shape[0] - 1
    d1 = pts[:i] - pts[i]
    for j in range(i):
        for k in range(j):
            if abs(d1[k, 0] * d1[j, 1] - d1[k, 1] * d1[j, 0]) <= 1.1920929e-07 * (abs(d1[j, 0]) + abs(d1[j, 1]) + abs(d1[k, 0]) + abs(d1[k, 1])):
                return True
    return False


def draw_subsets(rng: CvRNG, count: int, model_points: int, n_subsets: int, m1: Optional[np.ndarray] = None,
                 m2: Optional[np.ndarray] = None, max_attempts: int = 10000) -> np.ndarray:
    """n_subsets index sets of distinct indices drawn as RANSACPointSetRegistrator::getSubset does;
    with m1/m2 given, subsets whose last point is collinear with an earlier pair are redrawn."""
    out = np.empty((n_subsets, model_points), np.int64)
    for s in range(n_subsets):
        for _ in range(max_attempts):
            idx = []
            while len(idx) < model_points:
                j = rng.uniform(0, count)
                if j not in idx:
                    idx.append(j)
            if m1 is not None and (_collinear_with_last(m1[idx]) or _collinear_with_last(m2[idx])):
                continue
            break
        out[s] = idx
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


# ---------------------------------------------------------------- 7-point F
def seven_point(m1: np.ndarray, m2: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Batched 7-point algorithm.  m1, m2: (S,7,2).  -> (F (C,3,3) with F[2,2] = 1 where possible,
    owner (C,) sample index of each solution); 1..3 real solutions per sample."""
    S = m1.shape[0]
    x1, y1, x2, y2 = m1[..., 0], m1[..., 1], m2[..., 0], m2[..., 1]
    A = np.stack([x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, np.ones_like(x1)], axis=-1).astype(np.float64)
    _, _, Vt = np.linalg.svd(A)
    f1, f2 = Vt[:, 7].reshape(S, 3, 3), Vt[:, 8].reshape(S, 3, 3)
    # det(l f1 + (1-l) f2) is a cubic in l: fit it through four evaluations
    ls = np.array([0.0, 1.0, -1.0, 2.0])
    dets = np.stack([np.linalg.det(l * f1 + (1 - l) * f2) for l in ls], axis=-1)
    V = np.vander(ls, 4)                       # columns l^3 l^2 l 1
    coef = np.linalg.solve(V, dets.T).T        # (S,4): c3 c2 c1 c0
    Fs, owner = [], []
    for s in range(S):
        c = coef[s]
        if abs(c[0]) < 1e-300:
            roots = np.roots(c[1:]) if np.any(np.abs(c[1:]) > 0) else np.array([])
        else:
            roots = np.roots(c)
        for r in roots:
            if abs(r.imag) > 1e-10 * max(1.0, abs(r.real)):
                continue
            F = r.real * f1[s] + (1 - r.real) * f2[s]
            if abs(F[2, 2]) > np.finfo(np.float64).eps:
                F = F / F[2, 2]
            Fs.append(F)
            owner.append(s)
    if not Fs:
        return np.zeros((0, 3, 3)), np.zeros(0, np.int64)
    return np.stack(Fs), np.asarray(owner, np.int64)


def find_fundamental_mat(pts0: np.ndarray, pts1: np.ndarray, thr: float, conf: float, max_iters: int,
                         device: int = 0) -> Tuple[Optional[np.ndarray], np.ndarray]:
    """-> (F (3,3) float64 or None, mask (n,1) uint8), as cv2.findFundamentalMat(..., FM_RANSAC, ...)."""
    m1 = np.ascontiguousarray(pts0, np.float32).reshape(-1, 2)
    m2 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2)
    n = m1.shape[0]
    if m2.shape[0] != n:
        raise ValueError("find_fundamental_mat: point sets differ in length")
    if n < 8:  # OpenCV needs >= 7 and switches estimators below 15; keep the RANSAC form for n >= 8
        return None, np.zeros((n, 1), np.uint8)
    rng = CvRNG()
    subsets = draw_subsets(rng, n, 7, int(max_iters), m1, m2)
    Fs, owner = seven_point(m1[subsets], m2[subsets])
    if Fs.shape[0] == 0:
        return None, np.zeros((n, 1), np.uint8)
    ctx = _runtime.context(device)
    counts, _ = hip.score_fundamental(ctx, m1, m2, Fs, thr, want_masks=False)
    win = replay_ransac_loop(counts, owner, n, 7, conf, int(max_iters))
    if win < 0:
        return None, np.zeros((n, 1), np.uint8)
    _, mask = hip.score_fundamental(ctx, m1, m2, Fs[win:win + 1], thr, want_masks=True)
    return Fs[win].copy(), mask.reshape(n, 1)


# ---------------------------------------------------------------- P3P
def _absolute_orientation(P: np.ndarray, Q: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """R, t with Q = R P + t for three point pairs (rows), via orthonormal triads."""
    def triad(A):
        e1 = A[1] - A[0]
        e1 = e1 / np.linalg.norm(e1)
        e3 = np.cross(e1, A[2] - A[0])
        e3 = e3 / np.linalg.norm(e3)
        return np.stack([e1, np.cross(e3, e1), e3], axis=1)
    R = triad(Q) @ triad(P).T
    return R, Q[0] - R @ P[0]


def p3p(X3: np.ndarray, bearings: np.ndarray):
    """Grunert's P3P for one sample.  X3: (3,3) world points, bearings: (3,3) unit rays.
    -> list of (R, t) world->camera (up to 4)."""
    P1, P2, P3 = X3
    j1, j2, j3 = bearings
    a2, b2, c2 = np.sum((P2 - P3) ** 2), np.sum((P1 - P3) ** 2), np.sum((P1 - P2) ** 2)
    if min(a2, b2, c2) < 1e-24:
        return []
    ca, cb, cg = j2 @ j3, j1 @ j3, j1 @ j2
    q = (a2 - c2) / b2
    p = (a2 + c2) / b2
    A4 = (q - 1) ** 2 - 4 * c2 / b2 * ca ** 2
    A3 = 4 * (q * (1 - q) * cb - (1 - p) * ca * cg + 2 * c2 / b2 * ca ** 2 * cb)
    A2 = 2 * (q ** 2 - 1 + 2 * q ** 2 * cb ** 2 + 2 * (b2 - c2) / b2 * ca ** 2 - 4 * p * ca * cb * cg + 2 * (b2 - a2) / b2 * cg ** 2)
    A1 = 4 * (-q * (1 + q) * cb + 2 * a2 / b2 * cg ** 2 * cb - (1 - p) * ca * cg)
    A0 = (1 + q) ** 2 - 4 * a2 / b2 * cg ** 2
    coef = np.array([A4, A3, A2, A1, A0])
    if not np.all(np.isfinite(coef)) or np.all(coef == 0):
        return []
    sols = []
    for v in np.roots(coef):
        if abs(v.imag) > 1e-8 * max(1.0, abs(v.real)) or v.real <= 0:
            continue
        v = v.real
        den = 2 * (cg - v * ca)
        if abs(den) < 1e-14:
            continue
        u = ((q - 1) * v * v - 2 * q * cb * v + 1 + q) / den
        if u <= 0:
            continue
        d = 1 + v * v - 2 * v * cb
        if d <= 0:
            continue
        s1 = np.sqrt(b2 / d)
        Q = np.stack([s1 * j1, u * s1 * j2, v * s1 * j3])
        R, t = _absolute_orientation(X3, Q)
        if np.all(np.isfinite(R)) and np.all(np.isfinite(t)):
            sols.append((R, t))
    return sols


def refine_pose(ctx, X: np.ndarray, x: np.ndarray, K: np.ndarray, R: np.ndarray, t: np.ndarray, iters: int = 10):
    """Gauss-Newton on the reprojection error over the given correspondences (pose only); the
    6x6 normal equations come from the GPU BA kernel with the points held fixed."""
    from scipy.spatial.transform import Rotation
    n = X.shape[0]
    cam = np.hstack([Rotation.from_matrix(R).as_rotvec(), t]).astype(np.float64)
    ci = np.zeros(n, np.int64)
    pi = np.arange(n, dtype=np.int64)
    obs = x.astype(np.float64)
    pts = X.astype(np.float64).ravel()
    prev = np.inf
    for _ in range(iters):
        ne = hip.ba_normal_eq(ctx, np.hstack([cam, pts]), 1, n, ci, pi, obs, K, 0)
        if not np.isfinite(ne["cost"]) or ne["cost"] > prev * (1 + 1e-12):
            break
        H, g = ne["Hcc"][0], ne["g"][:6]
        try:
            step = np.linalg.solve(H + 1e-9 * np.trace(H) / 6 * np.eye(6), -g)
        except np.linalg.LinAlgError:
            break
        if prev - ne["cost"] < 1e-10 * max(ne["cost"], 1e-300) and prev < np.inf:
            break
        prev = ne["cost"]
        cam = cam + step
    return Rotation.from_rotvec(cam[:3]).as_matrix(), cam[3:].copy()


def solve_pnp_ransac(X: np.ndarray, x: np.ndarray, K: np.ndarray, thr: float, conf: float, max_iters: int,
                     device: int = 0):
    """-> (success, R (3,3), t (3,), inliers (k,1) int32), as cv.solvePnPRansac(flags=SOLVEPNP_P3P)
    followed by its refit on the inliers (OpenCV refits with EPnP; here Gauss-Newton on the same error)."""
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
    Kinv = np.linalg.inv(Kd)
    rays = (Kinv @ np.hstack([x.astype(np.float64), np.ones((n, 1))]).T).T
    rays /= np.linalg.norm(rays, axis=1, keepdims=True)
    Xd = X.astype(np.float64)
    cands, owner = [], []
    for s, idx in enumerate(subsets):
        best, best_e = None, np.inf
        for R, t in p3p(Xd[idx[:3]], rays[idx[:3]]):
            pc = R @ Xd[idx[3]] + t
            if pc[2] <= 0:
                continue
            uv = Kd @ pc
            e = np.sum((uv[:2] / uv[2] - x[idx[3]]) ** 2)
            if e < best_e:
                best, best_e = (R, t), e
        if best is not None:
            cands.append(np.hstack([best[0], best[1][:, None]]))
            owner.append(s)
    if not cands:
        return False, np.eye(3), np.zeros(3), empty
    Rt = np.stack(cands)
    owner = np.asarray(owner, np.int64)
    ctx = _runtime.context(device)
    counts, _ = hip.score_pnp(ctx, X, x, Kd, Rt, thr, want_masks=False)
    win = replay_ransac_loop(counts, owner, n, 4, conf, int(max_iters))
    if win < 0:
        return False, np.eye(3), np.zeros(3), empty
    _, mask = hip.score_pnp(ctx, X, x, Kd, Rt[win:win + 1], thr, want_masks=True)
    inl = np.nonzero(mask[0])[0]
    R, t = Rt[win][:, :3], Rt[win][:, 3]
    if inl.size >= 4:
        R2, t2 = refine_pose(ctx, X[inl], x[inl], Kd, R, t)
        if np.all(np.isfinite(R2)) and np.all(np.isfinite(t2)):
            R, t = R2, t2
    return True, R, t, inl.astype(np.int32).reshape(-1, 1)
