"""vo_track_pnp_batch (the tracked frame of many streams in one launch chain, one synchronisation) against one
vo_track_pnp call per stream: tracked points, status, pose, inlier mask, counts and generator state must be
identical for every stream -- clean streams, streams with so many wrong landmarks that the RANSAC loop outlives the
first chunk of samples, streams with a handful of survivors and lost streams, all in one batch."""
import numpy as np
import pytest

from visual_odometry_amd import synth

pytestmark = pytest.mark.gpu

K = np.array([[420.0, 0, 160.0], [0, 420.0, 120.0], [0, 0, 1]])
SHAPE = (240, 320)


def _stream(seed, n, outlier_frac=0.0, lost=0, margin=25):
    """Frame pair with a known warp, keypoints, and landmarks that project onto the warped keypoints under a known
    pose; `lost` of the keypoints are pushed out of the image (they lose their status)."""
    img0, img1, A = synth.frame_pair(SHAPE, seed=seed, max_shift=3.0)
    pts = synth.uniform_keypoints(SHAPE, n, seed=seed + 1, margin=margin)
    rng = np.random.default_rng(seed + 2)
    moved = pts.astype(np.float64) @ A[:, :2].T + A[:, 2]
    R = synth.rodrigues([0.02, -0.03, 0.01])
    t = np.array([0.3, -0.1, 0.5])
    z = rng.uniform(4, 30, n)
    rays = np.linalg.solve(K, np.hstack([moved, np.ones((n, 1))]).T).T
    X = (R.T @ (rays * z[:, None] - t).T).T
    bad = rng.uniform(size=n) < outlier_frac
    X[bad] += rng.normal(0, 3.0, (int(bad.sum()), 3))
    pts = pts.copy()
    if lost:
        pts[n - lost:] += 5000.0
    return img0, img1, pts, X.astype(np.float32)


def _batch(specs, n):
    imgs0, imgs1, P, X = [], [], [], []
    for k, (outliers, lost) in enumerate(specs):
        a, b, p, x = _stream(100 + 7 * k, n, outliers, lost)
        imgs0.append(a); imgs1.append(b); P.append(p); X.append(x)
    return np.stack(imgs0), np.stack(imgs1), np.stack(P), np.stack(X)


def _singles(hip, ctx, imgs0, imgs1, P, X, thr, conf, iters, states=None):
    res = []
    h, w = imgs0.shape[1:]
    for b in range(len(P)):
        p0 = hip.Pyramid(ctx, 1, h, w, 3).build(imgs0[b])
        p1 = hip.Pyramid(ctx, 1, h, w, 3).build(imgs1[b])
        got = hip.track_pnp(ctx, p0, p1, P[b], X[b], K, thr, conf, iters, rng_state=None if states is None else int(states[b]))
        res.append(got + (hip.track_pnp_path(ctx),))
    return res


def _compare(batch, singles, with_path=True):
    pts, st, Rt, mask, tracked, good, states, path = batch
    for b, ref in enumerate(singles):
        np.testing.assert_array_equal(np.asarray(pts[b]), ref[0], err_msg=f"stream {b}: points")
        np.testing.assert_array_equal(np.asarray(st[b]), ref[1], err_msg=f"stream {b}: status")
        assert (int(tracked[b]), int(good[b])) == (ref[4], ref[5]), (b, tracked[b], good[b], ref[4:6])
        np.testing.assert_array_equal(np.asarray(mask[b]), ref[3], err_msg=f"stream {b}: mask")
        if ref[2] is None:
            assert good[b] == 0 and not Rt[b].any()
        else:
            np.testing.assert_array_equal(Rt[b], ref[2], err_msg=f"stream {b}: pose")
        assert int(states[b]) == ref[6], (b, int(states[b]), ref[6])
        if with_path:
            assert int(path[b]) == ref[7], (b, int(path[b]), ref[7])


# 64 streams: clean ones, every fourth with half its landmarks wrong (loop ends inside the first chunk, but late),
# some with three quarters wrong (the long loop), one nearly lost, one lost, one with four survivors
def _mixed_specs(n):
    specs = []
    for b in range(64):
        if b % 16 == 5:
            specs.append((0.75, 0))
        elif b % 4 == 1:
            specs.append((0.5, 0))
        else:
            specs.append((0.0 if b % 2 else 0.1, 0))
    specs[10] = (0.0, n - 6)
    specs[20] = (0.0, n)
    specs[30] = (0.0, n - 4)
    specs[40] = (0.0, n - 3)
    return specs


@pytest.mark.parametrize("stage1", ["32", "0"])
def test_64_streams_equal_64_single_calls(ctx, monkeypatch, stage1):
    from visual_odometry_amd import hip
    monkeypatch.setenv("VO_KLT_VARIANT", "3")          # the batch's size picks this tracker kernel; the single calls follow
    monkeypatch.setenv("VO_FRAME_STAGE1", stage1)      # staged first chunk (default) and the one-stage form
    n = 300
    imgs0, imgs1, P, X = _batch(_mixed_specs(n), n)
    p0 = hip.Pyramid(ctx, 64, *SHAPE, 3).build(imgs0)
    p1 = hip.Pyramid(ctx, 64, *SHAPE, 3).build(imgs1)
    got = hip.track_pnp_batch(ctx, p0, p1, P, X, K, 1.0, 0.999, 3000)
    ref = _singles(hip, ctx, imgs0, imgs1, P, X, 1.0, 0.999, 3000)
    _compare(got, ref)
    paths = got[7]
    assert (paths == 1).sum() >= 40 and (paths == 2).sum() >= 3 and paths[20] == 0 and paths[40] == 0
    assert got[5][0] > 200 and got[5][20] == 0


def test_per_stream_generator_states_and_device_arrays(ctx, monkeypatch):
    """Every stream from its own cv::RNG state, arrays resident on the GPU; a second call reuses the buffers."""
    import torch
    from visual_odometry_amd import hip
    monkeypatch.setenv("VO_KLT_VARIANT", "3")
    n = 500
    specs = [(0.0, 0), (0.5, 0), (0.8, 0), (0.1, 0), (0.0, n - 5), (0.3, 0), (0.0, n), (0.6, 0)]
    imgs0, imgs1, P, X = _batch(specs, n)
    states = np.array([12345678901234567 + 977 * b for b in range(8)], np.uint64)
    states[3] = np.uint64(0xFFFFFFFFFFFFFFFF)
    p0 = hip.Pyramid(ctx, 8, *SHAPE, 3).build(imgs0)
    p1 = hip.Pyramid(ctx, 8, *SHAPE, 3).build(imgs1)
    ref = _singles(hip, ctx, imgs0, imgs1, P, X, 1.0, 0.999, 2000, states)
    for _ in range(2):
        got = hip.track_pnp_batch(ctx, p0, p1, torch.from_numpy(P).cuda(), torch.from_numpy(X).cuda(), K, 1.0, 0.999, 2000,
                                  rng_state=states)
        got = (got[0].cpu().numpy(), got[1].cpu().numpy(), got[2], got[3].cpu().numpy()) + got[4:]
        _compare(got, ref)
    host = hip.track_pnp_batch(ctx, p0, p1, P, X, K, 1.0, 0.999, 2000, rng_state=states)
    _compare(host, ref)


def test_default_tracker_choice_against_composed_calls(ctx):
    """Without forcing the tracker kernel: the batch equals vo_klt_track on the batched pyramids followed, per stream,
    by the status filter and vo_solve_pnp_ransac."""
    from visual_odometry_amd import hip
    n = 400
    specs = [(0.05 * (b % 5), 0) for b in range(24)]
    imgs0, imgs1, P, X = _batch(specs, n)
    p0 = hip.Pyramid(ctx, 24, *SHAPE, 3).build(imgs0)
    p1 = hip.Pyramid(ctx, 24, *SHAPE, 3).build(imgs1)
    got = hip.track_pnp_batch(ctx, p0, p1, P, X, K, 1.0, 0.999, 1000)
    out, st, _ = hip.klt_track(ctx, p0, p1, P)
    np.testing.assert_array_equal(got[0], out)
    np.testing.assert_array_equal(got[1], st)
    for b in range(24):
        keep = st[b] == 1
        Rt, mask, good, state = hip.solve_pnp_ransac(ctx, X[b][keep], out[b][keep], K, 1.0, 0.999, 1000)
        full = np.zeros(n, np.uint8)
        full[keep] = mask
        assert (int(got[4][b]), int(got[5][b]), int(got[6][b])) == (int(keep.sum()), good, state)
        np.testing.assert_array_equal(got[3][b], full)
        np.testing.assert_array_equal(got[2][b], Rt)


def test_two_contexts_overlap(ctx):
    """begin on two contexts before the first finish (how a caller overlaps the chains of two launch groups)."""
    from visual_odometry_amd import hip
    n = 300
    imgs0, imgs1, P, X = _batch([(0.1, 0)] * 6, n)
    ctx2 = hip.Context(0)
    res = []
    jobs = []
    for c, lo in ((ctx, 0), (ctx2, 3)):
        p0 = hip.Pyramid(c, 3, *SHAPE, 3).build(imgs0[lo:lo + 3])
        p1 = hip.Pyramid(c, 3, *SHAPE, 3).build(imgs1[lo:lo + 3])
        jobs.append((hip.TrackPnpBatch(c).begin(p0, p1, P[lo:lo + 3], X[lo:lo + 3], K, 1.0, 0.999, 500), p0, p1))
    with pytest.raises(ValueError):
        jobs[0][0].begin(jobs[0][1], jobs[0][2], P[:3], X[:3], K, 1.0, 0.999, 500)      # one batch per context in flight
    for job, _, _ in jobs:
        res.append(job.finish())
    whole0 = hip.Pyramid(ctx, 6, *SHAPE, 3).build(imgs0)
    whole1 = hip.Pyramid(ctx, 6, *SHAPE, 3).build(imgs1)
    whole = hip.track_pnp_batch(ctx, whole0, whole1, P, X, K, 1.0, 0.999, 500)
    for k in range(7):
        np.testing.assert_array_equal(np.concatenate([res[0][k], res[1][k]]), whole[k])
