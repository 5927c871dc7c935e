"""vo_track_pnp (tracker, status filter and PnP-RANSAC in one call with one synchronisation) against the same
computation spelled out over the separate calls: vo_klt_track, the status == 1 filter on the host,
vo_solve_pnp_ransac.  The parts themselves are held to the CPU oracle elsewhere; here every output must be
identical -- tracked points, status, pose, inlier mask, counts and the generator state -- on the path that ends
after the first chunk of samples, on the one that needs more, and on the degenerate ones."""
import numpy as np
import pytest

from visual_odometry_amd import synth

pytestmark = pytest.mark.gpu

K = np.array([[420.0, 0, 160.0], [0, 420.0, 120.0], [0, 0, 1]])


def _scene(shape=(240, 320), n=600, seed=1, outlier_frac=0.0, margin=25):
    """Frame pair with a known warp, keypoints, and landmarks that project onto the warped keypoints under a known pose."""
    img0, img1, A = synth.frame_pair(shape, seed=seed, max_shift=3.0)
    pts = synth.uniform_keypoints(shape, n, seed=seed + 1, margin=margin)
    rng = np.random.default_rng(seed + 2)
    moved = pts.astype(np.float64) @ A[:, :2].T + A[:, 2]
    R = synth.rodrigues([0.02, -0.03, 0.01])
    t = np.array([0.3, -0.1, 0.5])
    z = rng.uniform(4, 30, n)
    rays = np.linalg.solve(K, np.hstack([moved, np.ones((n, 1))]).T).T
    X = (R.T @ (rays * z[:, None] - t).T).T
    bad = rng.uniform(size=n) < outlier_frac
    X[bad] += rng.normal(0, 3.0, (int(bad.sum()), 3))
    return img0, img1, pts, X.astype(np.float32), R, t


def _composed(hip, ctx, p0, p1, pts, X, thr, conf, iters):
    out, st, _ = hip.klt_track(ctx, p0, p1, pts)
    keep = st[0] == 1
    Rt, mask, good, state = hip.solve_pnp_ransac(ctx, X[keep], out[0][keep], K, thr, conf, iters)
    full = np.zeros(len(pts), np.uint8)
    if good > 0:
        full[keep] = mask
    return out[0], st[0], Rt, full, int(keep.sum()), good, state


def _check(hip, ctx, img0, img1, pts, X, thr=1.0, conf=0.999, iters=2000):
    h, w = img0.shape
    p0 = hip.Pyramid(ctx, 1, h, w, 3).build(img0)
    p1 = hip.Pyramid(ctx, 1, h, w, 3).build(img1)
    ref = _composed(hip, ctx, p0, p1, pts, X, thr, conf, iters)
    got = hip.track_pnp(ctx, p0, p1, pts, X, K, thr, conf, iters)
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])
    assert got[4] == ref[4] and got[5] == ref[5], (got[4:6], ref[4:6])
    np.testing.assert_array_equal(got[3], ref[3])
    if ref[2] is None:
        assert got[2] is None
    else:
        np.testing.assert_array_equal(got[2], ref[2])
    assert got[6] == ref[6]
    # the new frame's pyramid built inside the call from the image
    p2 = hip.Pyramid(ctx, 1, h, w, 3)
    inside = hip.track_pnp(ctx, p0, p2, pts, X, K, thr, conf, iters, next_image=img1)
    for a, b in zip(inside, got):
        if a is None or b is None:
            assert a is None and b is None
        else:
            np.testing.assert_array_equal(a, b)
    for lvl in range(p1.max_level + 1):
        np.testing.assert_array_equal(p2.download(0, lvl), p1.download(0, lvl))
    return got, (p0, p1)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_clean_scene_ends_after_the_first_chunk(ctx, seed):
    from visual_odometry_amd import hip
    img0, img1, pts, X, R, t = _scene(seed=seed)
    got, _ = _check(hip, ctx, img0, img1, pts, X)
    assert got[4] > 500 and got[5] > 0.9 * got[4]
    assert hip.track_pnp_path(ctx) == 1
    np.testing.assert_allclose(got[2][:, :3], R, atol=2e-3)
    np.testing.assert_allclose(got[2][:, 3], t, atol=3e-2)


@pytest.mark.parametrize("outliers,iters,path", [(0.5, 2000, 1), (0.75, 5000, 2), (0.5, 100, 1), (0.9, 300, 2)])
def test_outliers_take_the_long_loop(ctx, outliers, iters, path):
    """Half the landmarks wrong: the adaptive bound (~107 iterations) still falls inside the first 256 samples;
    three quarters wrong (~1 800) or an iteration cap above 256 that the bound never undercuts: the call goes on
    with more samples; a cap below 256 always ends on the first chunk."""
    from visual_odometry_amd import hip
    img0, img1, pts, X, R, t = _scene(seed=7, n=900, outlier_frac=outliers)
    _check(hip, ctx, img0, img1, pts, X, iters=iters)
    assert hip.track_pnp_path(ctx) == path


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 6, 9, 17, 40])
def test_few_points(ctx, n):
    """Four to a dozen survivors: most draws repeat an index, the raw draws kept on the device run out and the
    call finishes through the host-side sampler; below four there is no pose."""
    from visual_odometry_amd import hip
    img0, img1, pts, X, R, t = _scene(seed=4, n=max(n, 1))
    pts, X = pts[:n], X[:n]
    if n == 0:
        h, w = img0.shape
        p0 = hip.Pyramid(ctx, 1, h, w, 3).build(img0)
        p1 = hip.Pyramid(ctx, 1, h, w, 3).build(img1)
        got = hip.track_pnp(ctx, p0, p1, pts, X, K, 1.0, 0.999, 100)
        assert got[0].shape == (0, 2) and got[2] is None and got[4] == 0 and got[5] == 0
        return
    got, _ = _check(hip, ctx, img0, img1, pts, X, iters=300)
    if n < 4:
        assert got[2] is None and not got[3].any() and hip.track_pnp_path(ctx) == 0
    else:
        assert hip.track_pnp_path(ctx) in (1, 3)
    if n == 4:
        assert hip.track_pnp_path(ctx) == 3         # 256 subsets of 4 out of 4 need ~2100 draws


def test_untracked_rows_and_lost_frame(ctx):
    """Keypoints outside the image lose their status: they drop out of the PnP problem and of the mask; with all
    of them lost there is no pose."""
    from visual_odometry_amd import hip
    img0, img1, pts, X, R, t = _scene(seed=5, n=400, margin=0)
    pts = pts.copy()
    pts[::3] += 1000.0
    got, _ = _check(hip, ctx, img0, img1, pts, X)
    assert not got[1][::3].any() and not got[3][::3].any() and got[4] < 300 and got[5] > 100
    lost = pts + 5000.0
    got, _ = _check(hip, ctx, img0, img1, lost, X)
    assert got[2] is None and got[4] == 0 and not got[3].any()


def test_device_resident_points(ctx):
    import torch
    from visual_odometry_amd import hip
    for outliers in (0.0, 0.6):
        img0, img1, pts, X, R, t = _scene(seed=11, n=700, outlier_frac=outliers)
        got, (p0, p1) = _check(hip, ctx, img0, img1, pts, X)
        dev = hip.track_pnp(ctx, p0, p1, torch.from_numpy(pts).cuda(), torch.from_numpy(X).cuda(), K, 1.0, 0.999, 2000)
        p2 = hip.Pyramid(ctx, 1, img1.shape[0], img1.shape[1], 3)
        dev2 = hip.track_pnp(ctx, p0, p2, torch.from_numpy(pts).cuda(), torch.from_numpy(X).cuda(), K, 1.0, 0.999, 2000,
                             next_image=torch.from_numpy(img1).cuda())
        np.testing.assert_array_equal(dev2[0].cpu().numpy(), got[0])
        np.testing.assert_array_equal(dev2[3].cpu().numpy(), got[3])
        np.testing.assert_array_equal(dev[0].cpu().numpy(), got[0])
        np.testing.assert_array_equal(dev[1].cpu().numpy(), got[1])
        np.testing.assert_array_equal(dev[3].cpu().numpy(), got[3])
        np.testing.assert_array_equal(dev[2], got[2])
        assert dev[4:] == got[4:]


def test_repeated_calls_and_other_seed(ctx):
    """Buffers and the device copy of the generator's raw draws are reused from call to call; another start
    state replaces them."""
    from visual_odometry_amd import hip
    img0, img1, pts, X, R, t = _scene(seed=13, n=500)
    h, w = img0.shape
    p0 = hip.Pyramid(ctx, 1, h, w, 3).build(img0)
    p1 = hip.Pyramid(ctx, 1, h, w, 3).build(img1)
    first = hip.track_pnp(ctx, p0, p1, pts, X, K, 1.0, 0.999, 1000)
    again = hip.track_pnp(ctx, p0, p1, pts, X, K, 1.0, 0.999, 1000)
    for a, b in zip(first, again):
        np.testing.assert_array_equal(a, b)
    out, st, _ = hip.klt_track(ctx, p0, p1, pts)
    keep = st[0] == 1
    for state in (12345678901234567, first[6]):
        Rt, mask, good, after = hip.solve_pnp_ransac(ctx, X[keep], out[0][keep], K, 1.0, 0.999, 1000, rng_state=state)
        got = hip.track_pnp(ctx, p0, p1, pts, X, K, 1.0, 0.999, 1000, rng_state=state)
        np.testing.assert_array_equal(got[2], Rt)
        np.testing.assert_array_equal(got[3][keep], mask)
        assert got[5] == good and got[6] == after
