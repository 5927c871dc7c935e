"""CPU tests of the KLT oracle: independent cross-checks of its building blocks
(scipy.ndimage with mode='mirror' == BORDER_REFLECT_101) and known-answer flows.
cv2 is absent, so these pin the restatement against independent arithmetic and
analytic answers, not against OpenCV itself (parity unpinned)."""
import numpy as np
import pytest
from scipy import ndimage

from visual_odometry_amd import synth


def test_level_clamp(oracle):
    # after level l the next size is ((w+1)/2,(h+1)/2); stop when <= 21 in a dimension
    assert oracle.pyr_max_level(1241, 376, 8) == 4
    assert oracle.pyr_max_level(640, 480, 8) == 4
    assert oracle.pyr_max_level(1024, 768, 8) == 5
    assert oracle.pyr_max_level(800, 600, 8) == 4
    assert oracle.pyr_max_level(1241, 376, 3) == 3
    assert oracle.pyr_max_level(30, 30, 8) == 0


@pytest.mark.parametrize("shape", [(37, 53), (64, 64), (47, 156), (376, 1241)])
def test_pyr_down_vs_ndimage(oracle, shape):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    full = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.int64), k, axis=1, mode="mirror"), k, axis=0,
                               mode="mirror")
    ref = ((full[::2, ::2] + 128) >> 8).astype(np.uint8)
    got = oracle.pyr_down(img)
    assert got.shape == ((shape[0] + 1) // 2, (shape[1] + 1) // 2)
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("shape", [(24, 31), (94, 311)])
def test_scharr_vs_ndimage(oracle, shape):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    a = img.astype(np.int64)
    sm, df = np.array([3, 10, 3]), np.array([-1, 0, 1])
    dx = ndimage.correlate1d(ndimage.correlate1d(a, sm, axis=0, mode="mirror"), df, axis=1, mode="mirror")
    dy = ndimage.correlate1d(ndimage.correlate1d(a, df, axis=0, mode="mirror"), sm, axis=1, mode="mirror")
    got = oracle.scharr(img)
    np.testing.assert_array_equal(got[..., 0], dx)
    np.testing.assert_array_equal(got[..., 1], dy)


def test_klt_recovers_known_warp(oracle):
    img0, img1, A = synth.frame_pair((240, 320), seed=5, max_shift=5.0, noise=0.5)
    pts = synth.uniform_keypoints((240, 320), 200, seed=1, margin=30)
    out, st, err = oracle.klt(img0, img1, pts)
    truth = pts @ A[:, :2].T + A[:, 2]
    ok = st.ravel() == 1
    assert ok.mean() > 0.97
    d = np.linalg.norm(out[ok] - truth[ok], axis=1)
    assert np.median(d) < 0.05 and np.percentile(d, 95) < 0.2
    assert (err[ok] > 1e-2).all()


def test_klt_identity_and_flat(oracle):
    img0, _, _ = synth.frame_pair((120, 160), seed=3)
    pts = synth.uniform_keypoints((120, 160), 50, seed=2, margin=15)
    out, st, _ = oracle.klt(img0, img0, pts)
    assert (st == 1).all()
    np.testing.assert_allclose(out, pts, atol=1e-3)
    flat = np.full((120, 160), 90, np.uint8)
    out, st, err = oracle.klt(flat, flat, pts)
    assert (st == 0).all() and (err == 0).all()   # min eigenvalue 0 < threshold


def test_klt_outside_points(oracle):
    img0, img1, _ = synth.frame_pair((120, 160), seed=4)
    pts = np.array([[-40.0, 10.0], [500.0, 50.0], [5.0, 5.0], [80.0, -30.0], [159.0, 119.0]], np.float32)
    out, st, _ = oracle.klt(img0, img1, pts)
    assert st[0, 0] == 0 and st[1, 0] == 0 and st[3, 0] == 0


def _float_lk(img0, img1, pts, oracle, max_level=3, iters=10, eps=1e-3, win=21):
    """Pyramidal translational LK written independently of the oracle: float64 throughout, Scharr and
    bilinear sampling through scipy.ndimage (mirror = REFLECT_101 image border, zero derivative outside
    the image), the same Gauss-Newton update and stopping rules.  Only interior windows are meaningful
    (no status logic); returns positions."""
    half = (win - 1) / 2.0
    sm, df = np.array([3.0, 10.0, 3.0]), np.array([-1.0, 0.0, 1.0])
    ys, xs = np.mgrid[0:win, 0:win].astype(np.float64)
    out = np.zeros((pts.shape[0], 2))
    for n, p in enumerate(pts.astype(np.float64)):
        g = None
        for level in range(max_level, -1, -1):
            I = oracle.pyr_level(img0, level).astype(np.float64)
            J = oracle.pyr_level(img1, level).astype(np.float64)
            Ix = ndimage.correlate1d(ndimage.correlate1d(I, sm, axis=0, mode="mirror"), df, axis=1, mode="mirror")
            Iy = ndimage.correlate1d(ndimage.correlate1d(I, df, axis=0, mode="mirror"), sm, axis=1, mode="mirror")
            pl = p / (1 << level) - half
            q = pl.copy() if g is None else 2.0 * g - half
            samp = lambda img, o: ndimage.map_coordinates(img, [ys + o[1], xs + o[0]], order=1, mode="mirror")
            T, Tx, Ty = samp(I, pl) * 32.0, samp(Ix, pl), samp(Iy, pl)      # OpenCV keeps I at 5 extra bits
            A = np.array([[np.sum(Tx * Tx), np.sum(Tx * Ty)], [np.sum(Tx * Ty), np.sum(Ty * Ty)]])
            prev = None
            for j in range(iters):
                d = samp(J, q) * 32.0 - T
                b = np.array([np.sum(d * Tx), np.sum(d * Ty)])
                delta = -np.linalg.solve(A, b)
                q = q + delta
                if delta @ delta <= eps * eps:
                    break
                if prev is not None and abs(delta[0] + prev[0]) < 0.01 and abs(delta[1] + prev[1]) < 0.01:
                    q = q - delta * 0.5
                    break
                prev = delta
            g = q + half
        out[n] = g
    return out


def test_klt_against_independent_float_implementation(oracle):
    """The fixed-point restatement and a float64 LK built from scipy primitives land on the same points:
    what separates them is OpenCV's integer rounding (14-bit weights, int16 samples), below 1e-3 px."""
    img0, img1, _ = synth.frame_pair((240, 320), seed=9, max_shift=5.0, noise=0.5)
    pts = synth.uniform_keypoints((240, 320), 120, seed=3, margin=60)     # windows interior on all four levels
    got, st, _ = oracle.klt(img0, img1, pts, max_level=3)
    ref = _float_lk(img0, img1, pts, oracle, max_level=3)
    ok = st.ravel() == 1
    assert ok.mean() > 0.97
    d = np.abs(got[ok] - ref[ok]).max(axis=1)
    assert np.median(d) < 5e-4 and d.max() < 5e-3        # measured: median 7e-5 px, max 6e-4 px
