"""GPU parity of the pyramid + pyramidal LK kernels against the CPU oracle,
through the C-ABI.  Integer stages (pyramid) are bit-exact; tracked positions
must agree within 0.05 px with >= 99.5 % status agreement (float32 sums are
reduced in a different order on the GPU)."""
import numpy as np
import pytest

from visual_odometry_amd import synth

pytestmark = pytest.mark.gpu

POS_TOL = 0.05
STATUS_AGREE = 0.995


@pytest.fixture(params=["3", "1"], ids=["three-per-wave", "one-per-wave"])
def klt_kernel(request, monkeypatch):
    """Run a tracker test with each of the library's two LK kernels (the library reads the
    environment variable at every launch; three keypoints per wavefront is its default)."""
    monkeypatch.setenv("VO_KLT_VARIANT", request.param)
    return request.param


def _compare(oracle, ctx, img0, img1, pts, max_level=8, **kw):
    from visual_odometry_amd import hip
    h, w = img0.shape
    p0 = hip.Pyramid(ctx, 1, h, w, max_level).build(img0)
    p1 = hip.Pyramid(ctx, 1, h, w, max_level).build(img1)
    out, st, err = hip.klt_track(ctx, p0, p1, pts, **kw)
    ro, rs, re_ = oracle.klt(img0, img1, pts, max_level=max_level, **kw)
    out, st, err = out[0], st[0], err[0]
    rs, re_ = rs.ravel(), re_.ravel()
    agree = (st == rs).mean()
    both = (st == 1) & (rs == 1)
    d = np.abs(out[both] - ro[both]).max() if both.any() else 0.0
    return agree, d, both.sum(), (out, st, err, ro, rs, re_)


@pytest.mark.parametrize("shape", [(37, 53), (64, 64), (47, 156), (188, 621), (376, 1241), (480, 640)])
def test_pyramid_bit_exact(oracle, ctx, shape):
    from visual_odometry_amd import hip
    rng = np.random.default_rng(7)
    imgs = rng.integers(0, 256, (3,) + shape, dtype=np.uint8)
    pyr = hip.Pyramid(ctx, 3, shape[0], shape[1], 8).build(imgs)
    assert pyr.max_level == oracle.pyr_max_level(shape[1], shape[0], 8)
    for b in range(3):
        for l in range(pyr.max_level + 1):
            np.testing.assert_array_equal(pyr.download(b, l), oracle.pyr_level(imgs[b], l))


@pytest.mark.parametrize("batch,shape,max_level", [(1, (480, 640), 8), (1, (480, 640), 3), (2, (479, 637), 8), (4, (720, 1280), 8),
                                                   (1, (1080, 1920), 8), (1, (188, 621), 2), (1, (100, 130), 8), (5, (480, 640), 8)])
def test_pyramid_small_levels_in_one_launch(oracle, ctx, batch, shape, max_level):
    """The last two to four levels of up to four images are built by one launch (tiles of the last level, the levels
    in between recomputed per tile): every level bit-exact against pyrDown, for two, three and four such levels, odd
    sizes, a source level that is the caller's image (byte reads) or the pyramid's own (dword reads), host images;
    five images take the launch-per-level path and must agree just the same."""
    from visual_odometry_amd import hip
    rng = np.random.default_rng(batch * 1000 + shape[0])
    imgs = rng.integers(0, 256, (batch,) + shape, dtype=np.uint8)
    pyr = hip.Pyramid(ctx, batch, shape[0], shape[1], max_level).build(imgs if batch > 1 else imgs[0])
    assert pyr.max_level == oracle.pyr_max_level(shape[1], shape[0], max_level)
    for b in range(batch):
        for l in range(pyr.max_level + 1):
            np.testing.assert_array_equal(pyr.download(b, l), oracle.pyr_level(imgs[b], l))


@pytest.mark.parametrize("batch,shape", [(8, (50, 91)), (16, (44, 45)), (8, (376, 1241)), (5, (45, 44)), (1, (23, 300))])
def test_pyramid_batches_and_strides(oracle, ctx, batch, shape):
    """Batches that take the image-per-XCD workgroup order (multiples of 8) and those that do not, the narrowest
    levels, and device input whose rows and images are padded (strides larger than the image)."""
    import torch
    from visual_odometry_amd import hip
    rng = np.random.default_rng(batch)
    h, w = shape
    big = torch.from_numpy(rng.integers(0, 256, (batch, h + 3, w + 5), dtype=np.uint8)).cuda()
    view = big[:, 1:1 + h, 2:2 + w]                     # row stride w + 5, image stride (h + 3) * (w + 5)
    pyr = hip.Pyramid(ctx, batch, h, w, 8).build(view)
    ctx.sync()
    imgs = view.cpu().numpy()
    assert pyr.max_level == oracle.pyr_max_level(w, h, 8)
    for b in {0, batch // 2, batch - 1}:
        for l in range(pyr.max_level + 1):
            np.testing.assert_array_equal(pyr.download(b, l), oracle.pyr_level(imgs[b], l))


def test_pyramid_device_input(oracle, ctx):
    import torch
    from visual_odometry_amd import hip
    rng = np.random.default_rng(8)
    imgs = rng.integers(0, 256, (2, 100, 130), dtype=np.uint8)
    t = torch.from_numpy(imgs).cuda()
    pyr = hip.Pyramid(ctx, 2, 100, 130, 3).build(t)
    ctx.sync()
    for l in range(pyr.max_level + 1):
        np.testing.assert_array_equal(pyr.download(1, l), oracle.pyr_level(imgs[1], l))


@pytest.mark.parametrize("shape,n,seed", [((120, 160), 300, 1), ((240, 320), 500, 2), ((480, 640), 1000, 3)])
def test_klt_parity_small(oracle, ctx, klt_kernel, shape, n, seed):
    img0, img1, _ = synth.frame_pair(shape, seed=seed)
    pts = synth.uniform_keypoints(shape, n, seed=seed, margin=2)   # includes border windows
    agree, d, nb, _ = _compare(oracle, ctx, img0, img1, pts)
    assert nb > 0.8 * n
    assert agree >= STATUS_AGREE, agree
    assert d <= POS_TOL, d


def test_klt_parity_kitti_shape(oracle, ctx, klt_kernel):
    img0, img1, _ = synth.frame_pair(synth.KITTI_SHAPE, seed=2021)
    pts = synth.uniform_keypoints(synth.KITTI_SHAPE, 2000, seed=0, margin=11)
    for max_level in (3, 8):
        agree, d, nb, (out, st, err, ro, rs, re_) = _compare(oracle, ctx, img0, img1, pts, max_level=max_level)
        assert nb > 1800
        assert agree >= STATUS_AGREE, agree
        assert d <= POS_TOL, d
        np.testing.assert_allclose(err[st == 1], re_[st == 1], rtol=1e-4)


def test_klt_edge_points(oracle, ctx, klt_kernel):
    """Points outside the image, on the border, and in flat regions."""
    img0, img1, _ = synth.frame_pair((120, 160), seed=4)
    img0 = img0.copy(); img1 = img1.copy()
    img0[40:90, 60:120] = 77; img1[40:90, 60:120] = 77        # flat patch -> min-eigenvalue rejection
    pts = np.array([[-40.0, 10.0], [500.0, 50.0], [5.0, 5.0], [80.0, -30.0], [159.0, 119.0], [0.0, 0.0],
                    [90.0, 65.0], [85.5, 60.25], [158.7, 3.2], [1.5, 118.0], [30.0, 30.0]], np.float32)
    agree, d, nb, (out, st, err, ro, rs, re_) = _compare(oracle, ctx, img0, img1, pts)
    np.testing.assert_array_equal(st, rs)
    assert d <= POS_TOL
    assert st[0] == 0 and st[1] == 0 and st[6] == 0


def test_klt_batch_and_empty(oracle, ctx, klt_kernel):
    from visual_odometry_amd import hip
    B, shape, n = 16, (96, 128), 37        # batch multiple of 8 -> XCD-aware block mapping
    pairs = [synth.frame_pair(shape, seed=100 + b) for b in range(B)]
    i0 = np.stack([p[0] for p in pairs]); i1 = np.stack([p[1] for p in pairs])
    pts = np.stack([synth.uniform_keypoints(shape, n, seed=b, margin=5) for b in range(B)])
    p0 = hip.Pyramid(ctx, B, *shape).build(i0)
    p1 = hip.Pyramid(ctx, B, *shape).build(i1)
    out, st, err = hip.klt_track(ctx, p0, p1, pts)
    for b in (0, 7, 8, 15):
        ro, rs, _ = oracle.klt(i0[b], i1[b], pts[b])
        np.testing.assert_array_equal(st[b], rs.ravel())
        ok = rs.ravel() == 1
        assert np.abs(out[b][ok] - ro[ok]).max() <= POS_TOL
    # batch not a multiple of 8 -> linear mapping
    p0b = hip.Pyramid(ctx, 3, *shape).build(i0[:3]); p1b = hip.Pyramid(ctx, 3, *shape).build(i1[:3])
    out3, st3, _ = hip.klt_track(ctx, p0b, p1b, pts[:3])
    np.testing.assert_array_equal(st3, st[:3]); np.testing.assert_array_equal(out3, out[:3])
    # fewer keypoints than one wavefront / one workgroup holds
    for m in (1, 2, 4, 13):
        om, sm, _ = hip.klt_track(ctx, p0b, p1b, np.ascontiguousarray(pts[:3, :m]))
        np.testing.assert_array_equal(sm, st[:3, :m]); np.testing.assert_array_equal(om, out[:3, :m])
    # empty input
    o, s, e = hip.klt_track(ctx, p0b, p1b, np.zeros((3, 0, 2), np.float32))
    assert o.shape == (3, 0, 2) and s.shape == (3, 0)


def test_klt_device_buffers_match_host(ctx):
    import torch
    from visual_odometry_amd import hip
    shape, n = (200, 300), 256
    img0, img1, _ = synth.frame_pair(shape, seed=9)
    pts = synth.uniform_keypoints(shape, n, seed=3)
    p0 = hip.Pyramid(ctx, 1, *shape).build(img0); p1 = hip.Pyramid(ctx, 1, *shape).build(img1)
    out, st, err = hip.klt_track(ctx, p0, p1, pts)
    d0 = hip.Pyramid(ctx, 1, *shape).build(torch.from_numpy(img0).cuda())
    d1 = hip.Pyramid(ctx, 1, *shape).build(torch.from_numpy(img1).cuda())
    o, s, e = hip.klt_track(ctx, d0, d1, torch.from_numpy(pts).cuda())
    ctx.sync()
    np.testing.assert_array_equal(o.cpu().numpy()[0], out[0])
    np.testing.assert_array_equal(s.cpu().numpy()[0], st[0])


def test_klt_forward_backward(oracle, ctx, klt_kernel):
    from visual_odometry_amd import hip
    shape, n = (240, 320), 400
    img0, img1, A = synth.frame_pair(shape, seed=11)
    pts = synth.uniform_keypoints(shape, n, seed=4, margin=25)
    p0 = hip.Pyramid(ctx, 1, *shape).build(img0); p1 = hip.Pyramid(ctx, 1, *shape).build(img1)
    out, mask = hip.klt_track_fb(ctx, p0, p1, pts, lam=0.1)
    fwd, sf, _ = oracle.klt(img0, img1, pts)
    back, sb, _ = oracle.klt(img1, img0, fwd)
    ref = (sf.ravel() == 1) & (sb.ravel() == 1) & (np.abs(pts - back) < 0.1).all(1)
    assert (mask[0].astype(bool) == ref).mean() >= STATUS_AGREE
    both = ref & mask[0].astype(bool)
    assert both.sum() > 0.8 * n
    assert np.abs(out[0][both] - fwd[both]).max() <= POS_TOL
    assert (out[0][~mask[0].astype(bool)] == 0).all()


def test_bad_arguments(ctx):
    from visual_odometry_amd import hip
    with pytest.raises(ValueError):
        hip.Pyramid(ctx, 1, 10, 10)                    # smaller than the window
    p = hip.Pyramid(ctx, 1, 64, 64)
    q = hip.Pyramid(ctx, 1, 64, 64).build(np.zeros((64, 64), np.uint8))
    with pytest.raises(ValueError):
        hip.klt_track(ctx, p, q, np.zeros((4, 2), np.float32))   # p not built
    with pytest.raises(ValueError):
        p.build(np.zeros((32, 64), np.uint8))


def test_klt_negative_fourth_weight(oracle, ctx, klt_kernel):
    """Fractional offsets within ~3e-5 of an integer in one axis make all three rounded bilinear weights round
    up, so the fourth, 2^14 - iw00 - iw01 - iw10, is -1 (OpenCV keeps it as a signed int).  Such keypoints
    must track like their neighbours (a kernel that reads the weights as unsigned lanes sends them 100 px off)."""
    img0, img1, _ = synth.frame_pair((365, 615), seed=102, max_shift=7.0463036954220915)
    base = np.array([(380.49796, 220.00018), (100.12449, 60.000045), (200.00004, 150.3), (301.62, 99.99997),
                     (150.000045, 180.000045)], np.float32)
    pts = np.vstack([base, base * 2 % np.array([600, 350], np.float32) + 8, base + 0.25]).astype(np.float32)
    for ml in (0, 2, 8):
        agree, d, n_both, _ = _compare(oracle, ctx, img0, img1, pts, max_level=ml)
        assert agree == 1.0 and n_both >= 12 and d <= POS_TOL, (ml, agree, d)


def test_bench_workload_parity(oracle, ctx):
    """The benchmark's own workload: 64 KITTI-shaped streams of 2000 keypoints tracked in one launch from
    device-resident frames (4-level pyramid, maxLevel 3); three of the streams against the oracle."""
    import os
    import sys
    import torch
    from visual_odometry_amd import hip
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    distinct = 8                                             # eight seeded streams, repeated to fill the launch
    frames, pts, _ = bench.make_streams(list(range(distinct)), 2)
    S = 64
    f = np.tile(frames, (1, S // distinct, 1, 1))
    p = np.tile(pts, (S // distinct, 1, 1))
    tctx = hip.Context.from_torch(0)
    d0, d1 = torch.from_numpy(f[0].copy()).cuda(), torch.from_numpy(f[1].copy()).cuda()
    pyr0 = hip.Pyramid(tctx, S, bench.H, bench.W, bench.MAX_LEVEL).build(d0)
    pyr1 = hip.Pyramid(tctx, S, bench.H, bench.W, bench.MAX_LEVEL).build(d1)
    out, st, err = hip.klt_track(tctx, pyr0, pyr1, torch.from_numpy(p.copy()).cuda())
    tctx.sync()
    out, st = out.cpu().numpy(), st.cpu().numpy()
    # copies of one stream give identical results wherever they sit in the launch
    for s in range(distinct, S):
        assert (out[s] == out[s % distinct]).all() and (st[s] == st[s % distinct]).all()
    for s in (0, 3, 7):
        ro, rs, _ = oracle.klt(frames[0, s], frames[1, s], pts[s], max_level=bench.MAX_LEVEL)
        assert (st[s] == rs.ravel()).mean() >= STATUS_AGREE
        both = (st[s] == 1) & (rs.ravel() == 1)
        assert both.mean() > 0.99 and np.abs(out[s][both] - ro[both]).max() <= POS_TOL


@pytest.mark.parametrize("shape", [(22, 22), (22, 90), (44, 43), (88, 175), (176, 350)])
def test_klt_levels_of_22_pixels(oracle, ctx, klt_kernel, shape):
    """Image sizes whose last pyramid level is exactly 22 pixels wide or high, the smallest OpenCV's level rule
    admits (every level larger than the 21-pixel window); keypoints up to and beyond the border."""
    img0, img1, _ = synth.frame_pair(shape, seed=shape[0] + shape[1], max_shift=2.0)
    pts = synth.uniform_keypoints(shape, 400, seed=1, margin=-8)
    agree, d, n_both, _ = _compare(oracle, ctx, img0, img1, pts, max_level=8)
    assert agree >= STATUS_AGREE and d <= POS_TOL and n_both > 50


def test_klt_initial_estimates(oracle, ctx, klt_kernel):
    """init_pts (the search starts there instead of at prev_pts, OpenCV's OPTFLOW_USE_INITIAL_FLOW): a 40-pixel
    shift that the plain call cannot follow is tracked from rough initial estimates; GPU and oracle agree."""
    from visual_odometry_amd import hip
    h, w = 240, 320
    img0, img1, A = synth.frame_pair((h, w), seed=77, max_shift=40.0, margin=64)
    pts = synth.uniform_keypoints((h, w), 500, seed=5, margin=45)
    truth = (pts @ A[:, :2].T + A[:, 2]).astype(np.float32)
    init = (truth + np.random.default_rng(3).uniform(-3, 3, truth.shape)).astype(np.float32)
    p0 = hip.Pyramid(ctx, 1, h, w, 3).build(img0)
    p1 = hip.Pyramid(ctx, 1, h, w, 3).build(img1)
    out, st, _ = hip.klt_track(ctx, p0, p1, pts, init_pts=init)
    ro, rs, _ = oracle.klt(img0, img1, pts, max_level=3, init_pts=init)
    assert (st[0] == rs.ravel()).mean() >= STATUS_AGREE
    both = (st[0] == 1) & (rs.ravel() == 1)
    assert both.sum() > 300 and np.abs(out[0][both] - ro[both]).max() <= POS_TOL
    inside = both & (truth[:, 0] > 12) & (truth[:, 0] < w - 13) & (truth[:, 1] > 12) & (truth[:, 1] < h - 13)
    assert np.median(np.linalg.norm(out[0][inside] - truth[inside], axis=1)) < 0.1


@pytest.mark.parametrize("shape,levels", [((768, 1024), 6), ((600, 800), 5)], ids=["malaga-1024x768", "malaga-800x600"])
def test_klt_malaga_shapes_3000_keypoints(oracle, ctx, klt_kernel, shape, levels):
    """The RANSAC configuration's frame sizes (1024x768 as quoted; 800x600 is what the reference's Malaga loader
    reads, utils/loadData.py:79) with 3000 keypoints and the reference's maxLevel = 8, which the level-size rule
    clamps to 5 and 4."""
    from visual_odometry_amd import hip
    img0, img1, _ = synth.frame_pair(shape, seed=77, max_shift=8.0)
    pts = synth.uniform_keypoints(shape, 3000, seed=5, margin=4)
    assert hip.Pyramid(ctx, 1, shape[0], shape[1], 8).max_level == levels - 1 == oracle.pyr_max_level(shape[1], shape[0], 8)
    agree, d, n_both, _ = _compare(oracle, ctx, img0, img1, pts, max_level=8)
    assert n_both > 2500
    assert agree >= STATUS_AGREE and d <= POS_TOL, (agree, d)


def test_klt_translation_equivariance_full_size(ctx, klt_kernel):
    """A property that needs no checker and holds at the benchmark's full size: cropping both frames by a multiple of
    2^maxLevel pixels shifts every pyramid level by whole pixels, so keypoints far enough from the borders (the
    pyramid's and the window's reach at the coarsest level) see the same pixels in both runs and must come out at the
    same place, shifted by the crop -- up to the float32 rounding of coordinates that differ by the shift (observed
    5e-4 px on 0.5 % of the coordinates, where an iteration's stopping test falls the other way)."""
    import os
    import sys
    from visual_odometry_amd import hip
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    frames, _, _ = bench.make_streams([5], 2)
    img0, img1 = frames[0, 0], frames[1, 0]
    dy, dx = 16, 24                                   # multiples of 2^3
    H, W = img0.shape
    reach = 140                                       # > 2^3 * (window half 11 + flow + filter taps) pixels at level 0
    rng = np.random.default_rng(3)
    n = 2000
    p = np.stack([rng.uniform(reach + dx, W - reach, n), rng.uniform(reach + dy, H - reach, n)], axis=1).astype(np.float32)
    full = [hip.Pyramid(ctx, 1, H, W, bench.MAX_LEVEL).build(im) for im in (img0, img1)]
    out_a, st_a, err_a = hip.klt_track(ctx, full[0], full[1], p)
    crop = [hip.Pyramid(ctx, 1, H - dy, W - dx, bench.MAX_LEVEL).build(np.ascontiguousarray(im[dy:, dx:])) for im in (img0, img1)]
    out_b, st_b, err_b = hip.klt_track(ctx, crop[0], crop[1], p - np.float32([dx, dy]))
    assert (st_a[0] == st_b[0]).all() and st_a[0].mean() > 0.99
    np.testing.assert_allclose(out_a[0] - np.float32([dx, dy]), out_b[0], rtol=0, atol=5e-3)
    np.testing.assert_allclose(err_a[0], err_b[0], rtol=1e-3, atol=1e-6)      # min-eigenvalues: same template pixels
