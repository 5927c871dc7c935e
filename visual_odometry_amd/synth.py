"""Seeded synthetic inputs shaped like the benchmark configurations.

Frames: band-limited random texture (four octaves of Gaussian-filtered uniform
noise); a frame pair is the texture seen through a small translation + affine
warp with pixel noise.  Used by bench.py and the tests (no dataset ships).
"""
import numpy as np
from scipy import ndimage

KITTI_SHAPE = (376, 1241)
PARKING_SHAPE = (480, 640)
MALAGA_SHAPE = (768, 1024)

K_KITTI = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1]], dtype=np.float32)
K_MALAGA = np.array([[621.18428, 0, 404.0076], [0, 621.18428, 309.05989], [0, 0, 1]], dtype=np.float32)
K_PARKING = np.array([[331.37, 0, 320], [0, 369.568, 240], [0, 0, 1]], dtype=np.float32)


def texture(shape, rng, margin=32):
    """Float texture in [0,255], `margin` pixels larger than `shape` on every side."""
    h, w = shape[0] + 2 * margin, shape[1] + 2 * margin
    acc = np.zeros((h, w), np.float64)
    for sigma in (1.0, 2.0, 4.0, 8.0):
        n = ndimage.gaussian_filter(rng.uniform(-1, 1, (h, w)), sigma, mode="wrap")
        acc += n / n.std()
    acc -= acc.min()
    acc *= 255.0 / acc.max()
    return acc


def frame_pair(shape=KITTI_SHAPE, seed=2021, max_shift=6.0, max_affine=0.005, noise=1.0, margin=32):
    """-> (img0 u8, img1 u8, A (2,3) float64) with img1(x') = img0(x), x' = A @ [x, y, 1]."""
    rng = np.random.default_rng(seed)
    tex = texture(shape, rng, margin)
    h, w = shape
    img0 = tex[margin:margin + h, margin:margin + w]
    t = rng.uniform(-max_shift, max_shift, 2)
    L = np.eye(2) + rng.uniform(-max_affine, max_affine, (2, 2))
    c = np.array([(w - 1) / 2.0, (h - 1) / 2.0])
    # forward map x' = L (x - c) + c + t ; img1 samples img0 at the inverse map
    Linv = np.linalg.inv(L)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    sx = Linv[0, 0] * (xs - c[0] - t[0]) + Linv[0, 1] * (ys - c[1] - t[1]) + c[0]
    sy = Linv[1, 0] * (xs - c[0] - t[0]) + Linv[1, 1] * (ys - c[1] - t[1]) + c[1]
    img1 = ndimage.map_coordinates(tex, [sy + margin, sx + margin], order=1, mode="nearest")
    img1 = img1 + rng.normal(0, noise, img1.shape)
    A = np.hstack([L, (c + t - L @ c)[:, None]])
    to_u8 = lambda a: np.clip(np.rint(a), 0, 255).astype(np.uint8)
    return to_u8(img0), to_u8(img1), A


def uniform_keypoints(shape, n, seed=0, margin=11):
    rng = np.random.default_rng(seed)
    h, w = shape
    x = rng.uniform(margin, w - 1 - margin, n)
    y = rng.uniform(margin, h - 1 - margin, n)
    return np.stack([x, y], 1).astype(np.float32)


def rodrigues(rv):
    """Rotation vector -> matrix (float64)."""
    rv = np.asarray(rv, np.float64)
    th = np.linalg.norm(rv)
    if th == 0:
        return np.eye(3)
    k = rv / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx


def two_view_scene(n=3000, K=K_MALAGA, shape=MALAGA_SHAPE, inlier_frac=0.7, noise=0.3, seed=7):
    """Seeded two-view scene -> dict(m1, m2 (n,2) f32, X (n,3) f32 world points, R, t, F_true, inlier mask)."""
    rng = np.random.default_rng(seed)
    Kd = K.astype(np.float64)
    h, w = shape
    # points sampled through camera 1's pixels at random depth -> all visible in image 1
    uv = np.stack([rng.uniform(0, w - 1, n), rng.uniform(0, h - 1, n)], 1)
    z = rng.uniform(6, 60, n)
    X = (np.linalg.inv(Kd) @ np.hstack([uv, np.ones((n, 1))]).T).T * z[:, None]
    R = rodrigues([0.01, -0.04, 0.008])
    t = np.array([0.6, -0.05, 0.9])
    x2 = (Kd @ (R @ X.T + t[:, None])).T
    x2 = x2[:, :2] / x2[:, 2:]
    inl = rng.uniform(size=n) < inlier_frac
    m1 = uv + rng.normal(0, noise, uv.shape) * inl[:, None]
    m2 = np.where(inl[:, None], x2 + rng.normal(0, noise, x2.shape),
                  np.stack([rng.uniform(0, w - 1, n), rng.uniform(0, h - 1, n)], 1))
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ki = np.linalg.inv(Kd)
    F = Ki.T @ tx @ R @ Ki
    return dict(m1=m1.astype(np.float32), m2=m2.astype(np.float32), X=X.astype(np.float32), R=R, t=t,
                F=F / F[2, 2], inliers=inl, K=Kd)


def fundamental_hypotheses(scene, H, seed=7):
    """H seeded candidates: the true F, scaled and perturbed copies, and random matrices."""
    rng = np.random.default_rng(seed)
    F0 = scene["F"]
    out = np.empty((H, 3, 3))
    for h in range(H):
        k = h % 4
        if h == 0:
            out[h] = F0
        elif k == 0:
            out[h] = F0 * rng.uniform(0.1, 10)
        elif k == 1:
            out[h] = F0 * (1 + rng.normal(0, 10 ** rng.uniform(-6, -2), (3, 3)))
        elif k == 2:
            out[h] = F0 + rng.normal(0, 1, (3, 3)) * np.abs(F0).max() * 10 ** rng.uniform(-5, -1)
        else:
            out[h] = rng.normal(0, 1, (3, 3))
    return out


def pose_hypotheses(scene, H, seed=7):
    """H seeded [R|t] candidates around the true pose."""
    rng = np.random.default_rng(seed)
    out = np.empty((H, 3, 4))
    for h in range(H):
        s = 0.0 if h == 0 else 10 ** rng.uniform(-6, -1)
        R = rodrigues(rng.normal(0, s, 3)) @ scene["R"]
        t = scene["t"] + rng.normal(0, s, 3)
        out[h] = np.hstack([R, t[:, None]])
    return out


def ba_problem(F=10, P=5000, K=K_KITTI, visibility=1.0, seed=0, obs_noise=0.5):
    """Seeded windowed-BA problem: points in front of cameras advancing along z.
    -> dict(x (6F+3P) f64, ci, pi int64 (sorted by camera), obs (O,2) f32, K f64)."""
    rng = np.random.default_rng(seed)
    Kd = K.astype(np.float64)
    pts = rng.uniform([-20, -3, 6], [20, 3, 60], (P, 3))
    cams = np.zeros((F, 6))
    cams[:, :3] = rng.normal(0, 0.01, (F, 3))
    cams[:, 5] = -0.5 * np.arange(F)
    ci, pi, obs = [], [], []
    for f in range(F):
        R = rodrigues(cams[f, :3])
        Xc = pts @ R.T + cams[f, 3:]
        uv = (Xc @ Kd.T)
        uv = uv[:, :2] / uv[:, 2:]
        vis = rng.uniform(size=P) < visibility
        idx = np.nonzero(vis)[0]
        ci.append(np.full(idx.size, f, np.int64)); pi.append(idx.astype(np.int64))
        obs.append(uv[idx] + rng.normal(0, obs_noise, (idx.size, 2)))
    x = np.hstack([(cams + rng.normal(0, 1e-3, cams.shape)).ravel(), (pts + rng.normal(0, 0.05, pts.shape)).ravel()])
    return dict(x=x, ci=np.concatenate(ci), pi=np.concatenate(pi), obs=np.concatenate(obs).astype(np.float32), K=Kd,
                F=F, P=P)


def ba_window_problem(F=107, births_every=7, per_birth=1600, survive=0.86, K=K_KITTI, seed=0, obs_noise=0.4):
    """Seeded BA window shaped like the frame loop's: `per_birth` landmarks start at every `births_every`-th camera
    (a re-bootstrap) and each is seen by consecutive cameras from there on, surviving a frame with probability
    `survive` -- runs of neighbouring cameras, a few observations for most landmarks and dozens for some, where
    ba_problem spreads a landmark's observations over random cameras.  Landmarks are numbered in order of first
    appearance and the observations are sorted by camera, then landmark.  Same dict as ba_problem."""
    rng = np.random.default_rng(seed)
    Kd = K.astype(np.float64)
    cams = np.zeros((F, 6))
    cams[:, :3] = rng.normal(0, 0.01, (F, 3))
    cams[:, 5] = -0.5 * np.arange(F)
    births = np.repeat(np.arange(0, F, births_every), per_birth)
    P = births.size
    life = rng.geometric(1.0 - survive, P)                      # cameras that see the landmark, birth included
    last = np.minimum(births + life, F)                          # exclusive
    pts = rng.uniform([-20, -3, 6], [20, 3, 60], (P, 3))
    pts[:, 2] += 0.5 * births                                   # in front of the camera that first sees it
    ci, pi, obs = [], [], []
    for f in range(F):
        idx = np.nonzero((births <= f) & (f < last))[0]
        Xc = pts[idx] @ rodrigues(cams[f, :3]).T + cams[f, 3:]
        ok = Xc[:, 2] > 1.0
        idx, Xc = idx[ok], Xc[ok]
        uv = Xc @ Kd.T
        uv = uv[:, :2] / uv[:, 2:]
        ci.append(np.full(idx.size, f, np.int64)); pi.append(idx.astype(np.int64))
        obs.append(uv + rng.normal(0, obs_noise, (idx.size, 2)))
    x = np.hstack([(cams + rng.normal(0, 1e-3, cams.shape)).ravel(), (pts + rng.normal(0, 0.05, pts.shape)).ravel()])
    return dict(x=x, ci=np.concatenate(ci), pi=np.concatenate(pi), obs=np.concatenate(obs).astype(np.float32), K=Kd,
                F=F, P=P)


def sift_like_descriptors(n1=3000, n2=3000, shared=0.6, jitter=6.0, seed=5):
    """Two sets of 128-entry descriptors shaped like SIFT's: non-negative, normalised to length 512,
    entries clipped and rounded to bytes, stored as float32.  The first `shared * min(n1, n2)` rows of
    the first set reappear (perturbed, at shuffled rows) in the second; the rest are unrelated.
    Returns (des1, des2, truth) with truth[q] = row of des2 that des1[q] was copied to, or -1."""
    rng = np.random.default_rng(seed)

    def finish(x):
        x = np.maximum(x, 0)
        x *= 512.0 / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-9)
        return np.minimum(np.rint(x), 255).astype(np.float32)

    base1 = rng.gamma(0.6, 1.0, (n1, 128))
    des1 = finish(base1.copy())
    m = int(shared * min(n1, n2))
    base2 = rng.gamma(0.6, 1.0, (n2, 128))
    perm = rng.permutation(n2)
    truth = np.full(n1, -1, np.int64)
    if m:
        base2[perm[:m]] = des1[:m] / 512.0 * np.linalg.norm(base1[:m], axis=1, keepdims=True)
        truth[:m] = perm[:m]
    des2 = finish(base2)
    if m:
        des2[perm[:m]] = np.clip(np.rint(des1[:m] + rng.normal(0, jitter, (m, 128))), 0, 255).astype(np.float32)
    return des1, des2, truth


class BoxScene:
    """Seeded piecewise-planar 3-D scene (back wall, ground, fronto-parallel box faces) seen by a
    camera that mostly translates sideways, in the manner of the 640x480 parking sequence.
    Frames are ray-cast: every pixel takes the nearest plane hit and samples that plane's
    band-limited texture bilinearly, so image motion has true parallax and occlusion edges.

    poses[i]: (4,4) float64 camera-to-world of frame i (frame 0 is the identity)."""

    PPU = 32.0   # texture pixels per world unit

    def __init__(self, n_frames=40, shape=PARKING_SHAPE, K=K_PARKING, seed=5, step=0.08, noise=1.0):
        rng = np.random.default_rng(seed)
        self.shape, self.K, self.noise = shape, np.asarray(K, np.float64), noise
        self.rng = rng
        x0, x1 = -14.0, 14.0 + n_frames * step
        # (origin, u axis, v axis, extent along u, extent along v); hit = origin + a u + b v
        self.planes = [
            (np.array([x0, -9.0, 14.0]), np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), x1 - x0, 11.0),   # back wall
            (np.array([x0, 1.6, 1.0]), np.array([1.0, 0, 0]), np.array([0, 0, 1.0]), x1 - x0, 13.0),     # ground
        ]
        x = x0 + 2.0
        while x < x1 - 4.0:
            wdt, hgt, z = rng.uniform(2.0, 4.0), rng.uniform(1.5, 3.0), rng.uniform(6.0, 10.5)
            self.planes.append((np.array([x, 1.6 - hgt, z]), np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), wdt, hgt))
            x += wdt + rng.uniform(1.0, 3.0)
        self.tex = []
        for (_, _, _, eu, ev) in self.planes:
            hw = (int(np.ceil(ev * self.PPU)) + 2, int(np.ceil(eu * self.PPU)) + 2)
            self.tex.append(texture(hw, rng, margin=0))
        self.poses = []
        for i in range(n_frames):
            T = np.eye(4)
            T[:3, :3] = rodrigues([0.004 * np.sin(0.21 * i), 0.010 * np.sin(0.13 * i), 0.003 * np.sin(0.17 * i)])
            T[:3, 3] = [step * i, 0.02 * np.sin(0.3 * i), 0.15 * step * i]
            self.poses.append(T)
        self.poses = np.asarray(self.poses)

    def render(self, i):
        h, w = self.shape
        T = self.poses[i]
        R, C = T[:3, :3], T[:3, 3]
        ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
        rays = (np.linalg.inv(self.K) @ np.stack([xs.ravel(), ys.ravel(), np.ones(h * w)])).T @ R.T
        best = np.full(h * w, np.inf)
        img = np.full(h * w, 128.0)
        for (o, u, v, eu, ev), tex in zip(self.planes, self.tex):
            nrm = np.cross(u, v)
            den = rays @ nrm
            with np.errstate(divide="ignore", invalid="ignore"):
                s = ((o - C) @ nrm) / den
                X = C + s[:, None] * rays - o
            a, b = X @ u, X @ v
            hit = (den != 0) & (s > 1e-6) & (s < best) & (a >= 0) & (a <= eu) & (b >= 0) & (b <= ev)
            if hit.any():
                img[hit] = ndimage.map_coordinates(tex, [b[hit] * self.PPU, a[hit] * self.PPU], order=1, mode="nearest")
                best[hit] = s[hit]
        img = img.reshape(h, w) + np.random.default_rng(1000 + i).normal(0, self.noise, (h, w))
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)

    def frames(self):
        """Iterator of (K float32, image uint8), the form the pipeline's Dataset carries."""
        Kf = np.asarray(self.K, np.float32)
        for i in range(len(self.poses)):
            yield Kf, self.render(i)
