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
