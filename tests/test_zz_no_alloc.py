"""vo_klt_track_fb_noalloc -- the forward-backward tracker for callers that cannot allocate -- against
vo_klt_track_fb through the C-ABI: identical points and masks (same kernels, deterministic), on two calls in
a row, the second with a smaller input; sizes beyond the fixed scratch are refused; the call leaves the
amount of free device memory untouched."""
import ctypes

import numpy as np
import pytest

from visual_odometry_amd import _lib, hip, synth

pytestmark = pytest.mark.gpu


def _fb_noalloc(ctx, prev, nxt, pts, lam, out, mask, max_iters=10, eps=0.001, min_eig=1e-2):
    prm = _lib.KltParams(max_iters, eps, min_eig)
    n = pts.shape[1]
    return ctx.lib.vo_klt_track_fb_noalloc(ctx.handle, prev.handle, nxt.handle, ctypes.c_void_p(pts.data_ptr()), n,
                                           ctypes.byref(prm), ctypes.c_float(lam), ctypes.c_void_p(out.data_ptr()),
                                           ctypes.c_void_p(mask.data_ptr()))


def test_noalloc_matches_track_fb_twice(ctx):
    import torch
    dev = torch.device("cuda", 0)
    lam = 0.1
    # first call: 6 images x 3000 keypoints (the throughput kernel); second: 2 images x 500 (the latency kernel)
    for batch, n, shape, seed in [(6, 3000, (240, 320), 3), (2, 500, (120, 160), 4)]:
        pairs = [synth.frame_pair(shape, seed=seed + b) for b in range(batch)]
        img0 = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev)
        img1 = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)
        pts = torch.from_numpy(np.stack([synth.uniform_keypoints(shape, n, seed=seed + b, margin=3)
                                         for b in range(batch)])).to(dev)
        p0 = hip.Pyramid(ctx, batch, *shape).build(img0)
        p1 = hip.Pyramid(ctx, batch, *shape).build(img1)
        ref_pts, ref_mask = hip.klt_track_fb(ctx, p0, p1, pts, lam)
        ctx.sync()
        out = torch.full_like(pts, -7.0)
        mask = torch.full((batch, n), 9, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(0)[0]
        rc = _fb_noalloc(ctx, p0, p1, pts, lam, out, mask)
        assert rc == 0, ctx.lib.vo_last_error()
        ctx.sync()
        assert torch.cuda.mem_get_info(0)[0] == free_before          # nothing allocated, nothing freed
        assert 0 < int(ref_mask.sum()) < batch * n                    # the comparison sees both outcomes
        assert torch.equal(mask, ref_mask)
        assert torch.equal(out, ref_pts)


def test_noalloc_refuses_sizes_beyond_its_scratch(ctx):
    import torch
    dev = torch.device("cuda", 0)
    shape = (64, 64)
    img = torch.zeros((1,) + shape, dtype=torch.uint8, device=dev)
    p0 = hip.Pyramid(ctx, 1, *shape).build(img)
    n = 262144 + 1
    pts = torch.zeros((1, n, 2), dtype=torch.float32, device=dev)
    out = torch.zeros_like(pts)
    mask = torch.zeros((1, n), dtype=torch.uint8, device=dev)
    rc = _fb_noalloc(ctx, p0, p0, pts, 0.1, out, mask)
    assert rc == -1 and b"exceed the fixed scratch" in ctx.lib.vo_last_error()
    assert _fb_noalloc(ctx, p0, p0, pts[:, :0], 0.1, out, mask) == 0         # empty input is a no-op


def test_noalloc_refuses_an_overlapping_call_from_another_stream(ctx):
    """All calls on a GPU share the scratch array: a call on another stream while the previous call's kernels are
    still running is refused; once that stream has been waited for, it goes through with the same result."""
    import torch
    dev = torch.device("cuda", 0)
    shape, batch, n = (376, 1241), 16, 4000
    pairs = [synth.frame_pair(shape, seed=40 + b % 2) for b in range(2)]
    img0 = torch.from_numpy(np.stack([pairs[b % 2][0] for b in range(batch)])).to(dev)
    img1 = torch.from_numpy(np.stack([pairs[b % 2][1] for b in range(batch)])).to(dev)
    pts = torch.from_numpy(np.stack([synth.uniform_keypoints(shape, n, seed=b, margin=3) for b in range(batch)])).to(dev)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s1):
        c1 = hip.Context.from_torch(0)
    with torch.cuda.stream(s2):
        c2 = hip.Context.from_torch(0)
    torch.cuda.synchronize()
    p0, p1 = hip.Pyramid(c1, batch, *shape).build(img0), hip.Pyramid(c1, batch, *shape).build(img1)
    q0, q1 = hip.Pyramid(c2, batch, *shape).build(img0), hip.Pyramid(c2, batch, *shape).build(img1)
    torch.cuda.synchronize()
    out1, out2 = torch.empty_like(pts), torch.empty_like(pts)
    m1 = torch.empty((batch, n), dtype=torch.uint8, device=dev)
    m2 = torch.empty_like(m1)
    assert _fb_noalloc(c1, p0, p1, pts, 0.1, out1, m1) == 0              # ~0.3 ms of kernels queued on stream 1
    assert _fb_noalloc(c1, p0, p1, pts, 0.1, out1, m1) == 0              # the same stream again: ordered, accepted
    rc = _fb_noalloc(c2, q0, q1, pts, 0.1, out2, m2)                      # stream 2 while stream 1 still runs
    assert rc == -4 and b"still in use" in c2.lib.vo_last_error()
    c1.sync()
    assert _fb_noalloc(c2, q0, q1, pts, 0.1, out2, m2) == 0, c2.lib.vo_last_error()
    c2.sync()
    assert torch.equal(out1, out2) and torch.equal(m1, m2)
    for o in (p0, p1, q0, q1):
        o.close()
    c1.close(); c2.close()
