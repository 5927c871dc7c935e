"""vo_harris_mask and vo_corner_subpix (through the C-ABI) against the CPU oracle, bit for bit, and the
FeatureExtractor.harris mirror built on them against known corner positions."""
import numpy as np
import pytest

from visual_odometry_amd import synth

pytestmark = pytest.mark.gpu


def _checker(h=120, w=160, cell=20, lo=50, hi=200, blur=1.0, seed=0):
    from scipy import ndimage
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.where(((yy // cell) + (xx // cell)) % 2 == 0, lo, hi).astype(np.float64)
    img = ndimage.gaussian_filter(img, blur) + np.random.default_rng(seed).normal(0, 0.5, (h, w))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("shape,seed", [(synth.KITTI_SHAPE, 2021), ((97, 131), 3), ((3, 3), 1)])
def test_mask_bit_exact(ctx, oracle, shape, seed):
    from visual_odometry_amd import hip
    img = synth.frame_pair(shape, seed=seed)[0] if shape[0] > 8 else np.random.default_rng(seed).integers(0, 255, shape).astype(np.uint8)
    for mode in (hip.HARRIS, hip.MIN_EIGEN):
        got = hip.harris_mask(ctx, img, 0.1, 0.01, mode)
        ref = oracle.harris_mask(img, 0.1, 0.01, mode)
        assert got.dtype == np.uint8 and set(np.unique(got)) <= {0, 255}
        assert (got == ref).all()
    batch = np.stack([img, img[::-1].copy()])
    got = hip.harris_mask(ctx, batch, 0.1, 0.01)
    assert (got[0] == oracle.harris_mask(batch[0])).all() and (got[1] == oracle.harris_mask(batch[1])).all()


def test_mask_device_resident(ctx, oracle):
    import torch
    from visual_odometry_amd import hip
    img = synth.frame_pair((200, 333), seed=4)[0]
    tctx = hip.Context.from_torch(0)
    got = hip.harris_mask(tctx, torch.from_numpy(img).cuda(), 0.1, 0.01)
    tctx.sync()
    assert (got.cpu().numpy() == oracle.harris_mask(img)).all()


@pytest.mark.parametrize("win", [(5, 5), (3, 3), (6, 2), (1, 1)])
def test_subpix_bit_exact_and_accurate(ctx, oracle, win):
    from visual_odometry_amd import hip
    img = _checker()
    true = np.array([(x - 0.5, y - 0.5) for y in range(20, 120, 20) for x in range(20, 160, 20)], np.float32)
    rng = np.random.default_rng(1)
    start = (true + rng.uniform(-2, 2, true.shape)).astype(np.float32)
    # corners near and beyond the border, and in flat regions (singular system: left where they are)
    extra = np.array([(0.3, 0.2), (159.5, 119.9), (-3.0, 50.0), (10.0, 10.0), (80.0, 2.0), (158.0, 60.0)], np.float32)
    pts = np.vstack([start, extra])
    got = hip.corner_subpix(ctx, img, pts, win, 50, 0.001)
    ref = oracle.corner_subpix(img, pts, win, 50, 0.001)
    assert got.dtype == np.float32 and got.shape == pts.shape
    assert (got.view(np.uint32) == ref.view(np.uint32)).all()
    if win == (5, 5):
        err = np.linalg.norm(got[:len(true)] - true, axis=1)
        assert np.median(err) < 0.05 and err.max() < 0.25
    assert (np.abs(got - pts) <= np.array(win) + 1e-6).all()      # a result beyond the window is discarded


def test_subpix_many_corners_and_edge_cases(ctx, oracle):
    import torch
    from visual_odometry_amd import hip
    img = synth.frame_pair((240, 320), seed=9)[0]
    pts = synth.uniform_keypoints((240, 320), 1000, seed=2, margin=0)
    got = hip.corner_subpix(ctx, img, pts)
    assert (got.view(np.uint32) == oracle.corner_subpix(img, pts).view(np.uint32)).all()
    assert hip.corner_subpix(ctx, img, np.zeros((0, 2), np.float32)).shape == (0, 2)
    tctx = hip.Context.from_torch(0)
    d = hip.corner_subpix(tctx, torch.from_numpy(img).cuda(), torch.from_numpy(pts).cuda(), (5, 5), 1, 0.0)
    tctx.sync()
    assert (d.cpu().numpy().view(np.uint32) == oracle.corner_subpix(img, pts, (5, 5), 1, 0.0).view(np.uint32)).all()
    with pytest.raises(Exception):
        hip.corner_subpix(ctx, img, pts, (7, 7))


def test_harris_mirror_on_checkerboard(ctx, oracle):
    """FeatureExtractor.harris: row 0 is the background centroid, the other rows are the refined corners."""
    from scipy import ndimage
    from visual_odometry_amd.vo_pipeline.featureExtraction import FeatureExtractor
    img = _checker()
    c = FeatureExtractor.harris(img)
    assert c.dtype == np.float32 and c.shape[1] == 2
    true = np.array([(x - 0.5, y - 0.5) for y in range(20, 120, 20) for x in range(20, 160, 20)], np.float32)
    assert c.shape[0] == 1 + len(true)
    d = np.linalg.norm(c[1:, None, :] - true[None], axis=2)
    assert (d.min(axis=1) < 0.25).all() and sorted(d.argmin(axis=1).tolist()) == list(range(len(true)))
    # same steps on the CPU: oracle mask -> components (first-block numbering) -> centroids -> oracle refinement
    assert (c == oracle.harris_corners(img)).all()
    # on this board blocks and pixels give the same order: components come row by row
    assert (np.diff(c[1:, 1].round(-1)) >= 0).all()
    m = FeatureExtractor.harris_maxima(img)
    assert m.shape[0] >= len(true) and m.dtype == np.float32
