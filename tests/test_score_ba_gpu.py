"""GPU parity through the C-ABI: hypothesis scoring must be bit-exact against the
CPU oracle (counts and masks); BA residual/cost/normal equations within 1e-6
relative on cost (in practice ~1e-13: float64, atomics reorder sums)."""
import os

import numpy as np
import pytest

from visual_odometry_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
COST_RTOL = 1e-6


@pytest.mark.parametrize("n,H,thr", [(3000, 2048, 0.1), (500, 33, 0.5), (1, 1, 0.1), (1025, 5, 0.1), (64, 300, 2.0)])
def test_score_fundamental_bit_exact(oracle, ctx, n, H, thr):
    from visual_odometry_amd import hip
    sc = synth.two_view_scene(n=n, seed=7)
    Fh = synth.fundamental_hypotheses(sc, H, seed=7)
    rc, rm = oracle.score_fundamental(sc["m1"], sc["m2"], Fh, thr)
    gc, gm = hip.score_fundamental(ctx, sc["m1"], sc["m2"], Fh, thr)
    np.testing.assert_array_equal(gc, rc)
    np.testing.assert_array_equal(gm, rm)
    gc2, gm2 = hip.score_fundamental(ctx, sc["m1"], sc["m2"], Fh, thr, want_masks=False)
    assert gm2 is None
    np.testing.assert_array_equal(gc2, rc)


@pytest.mark.parametrize("n,H,thr", [(3000, 2048, 1.0), (400, 17, 1.0), (2049, 3, 0.5)])
def test_score_pnp_bit_exact(oracle, ctx, n, H, thr):
    from visual_odometry_amd import hip
    sc = synth.two_view_scene(n=n, seed=9)
    Rt = synth.pose_hypotheses(sc, H, seed=9)
    rc, rm = oracle.score_pnp(sc["X"], sc["m2"], sc["K"], Rt, thr)
    gc, gm = hip.score_pnp(ctx, sc["X"], sc["m2"], sc["K"], Rt, thr)
    np.testing.assert_array_equal(gc, rc)
    np.testing.assert_array_equal(gm, rm)
    assert rc[0] > 0.4 * n


def test_score_degenerate_models(oracle, ctx):
    """Zero / NaN models and points behind the camera follow the same comparison rules."""
    from visual_odometry_amd import hip
    sc = synth.two_view_scene(n=200, seed=1)
    Fh = np.stack([np.zeros((3, 3)), np.full((3, 3), np.nan), sc["F"], np.eye(3)])
    rc, rm = oracle.score_fundamental(sc["m1"], sc["m2"], Fh, 0.1)
    gc, gm = hip.score_fundamental(ctx, sc["m1"], sc["m2"], Fh, 0.1)
    np.testing.assert_array_equal(gc, rc); np.testing.assert_array_equal(gm, rm)
    Rt = np.stack([np.zeros((3, 4)), np.hstack([np.eye(3), [[0], [0], [-100.0]]]), np.hstack([sc["R"], sc["t"][:, None]])])
    rc, rm = oracle.score_pnp(sc["X"], sc["m2"], sc["K"], Rt, 1.0)
    gc, gm = hip.score_pnp(ctx, sc["X"], sc["m2"], sc["K"], Rt, 1.0)
    np.testing.assert_array_equal(gc, rc); np.testing.assert_array_equal(gm, rm)


def test_score_device_buffers_and_empty(oracle, ctx):
    import torch
    from visual_odometry_amd import hip
    sc = synth.two_view_scene(n=777, seed=2)
    Fh = synth.fundamental_hypotheses(sc, 40, seed=2)
    rc, rm = oracle.score_fundamental(sc["m1"], sc["m2"], Fh, 0.1)
    gc, gm = hip.score_fundamental(ctx, torch.from_numpy(sc["m1"]).cuda(), torch.from_numpy(sc["m2"]).cuda(),
                                   torch.from_numpy(Fh).cuda(), 0.1)
    ctx.sync()
    np.testing.assert_array_equal(gc.cpu().numpy(), rc); np.testing.assert_array_equal(gm.cpu().numpy(), rm)
    c, m = hip.score_fundamental(ctx, np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), Fh, 0.1)
    assert (c == 0).all() and m.shape == (40, 0)
    c, m = hip.score_fundamental(ctx, sc["m1"], sc["m2"], np.zeros((0, 3, 3)), 0.1)
    assert c.shape == (0,)
    with pytest.raises(ValueError):
        hip.score_fundamental(ctx, sc["m1"], sc["m2"][:10], Fh, 0.1)


@pytest.fixture(params=["atomics", "gather"])
def landmark_path(request, monkeypatch):
    """vo_ba_normal_eq sums the landmark blocks with float64 atomics (small calls) or gathers them from
    landmark-major lists built on the device (large calls); the environment variable moves the switch point so
    that every case runs through both."""
    monkeypatch.setenv("VO_BA_GATHER_MIN_OBS", "0" if request.param == "gather" else "1000000000")
    return request.param


@pytest.mark.parametrize("name", ["ba_small", "ba_mid", "ba_frozen8"])
def test_ba_matches_golden_and_oracle(oracle, ctx, name, landmark_path):
    from visual_odometry_amd import hip
    d = np.load(os.path.join(GOLD, name + ".npz"))
    F, P, ci, pi = int(d["F"]), int(d["P"]), d["ci"], d["pi"]
    r, cost = hip.ba_residual(ctx, d["x"], F, P, ci, pi, d["obs"], d["K"])
    np.testing.assert_allclose(r, d["r"], rtol=0, atol=1e-8)                # reference _fun output
    assert abs(cost - float(d["cost"])) <= COST_RTOL * float(d["cost"])
    start = int(np.count_nonzero(ci < int(d["n_fixed"])))
    ne = hip.ba_normal_eq(ctx, d["x"], F, P, ci, pi, d["obs"], d["K"], start)
    ref = oracle.ba_normal_eq(d["x"], F, P, ci, pi, d["obs"], d["K"], start)
    assert abs(ne["cost"] - ref["cost"]) <= 1e-12 * ref["cost"]
    np.testing.assert_allclose(ne["r"], ref["r"], rtol=0, atol=1e-9)
    for k in ("Hcc", "Hpp", "Hcp", "g"):
        scale = np.abs(ref[k]).max()
        assert np.abs(ne[k] - ref[k]).max() <= 1e-10 * scale, k
    assert (ne["Hcp"][:start] == 0).all()


def test_ba_config4_shape(oracle, ctx, landmark_path):
    """10 keyframes x 5000 landmarks, all visible (50 000 observations), 3 frozen cameras."""
    from visual_odometry_amd import hip
    pb = synth.ba_problem(F=10, P=5000, seed=0)
    F, P = pb["F"], pb["P"]
    start = int(np.count_nonzero(pb["ci"] < int(F / 4) + 1))
    assert pb["ci"].size == 50000 and start == 15000
    ne = hip.ba_normal_eq(ctx, pb["x"], F, P, pb["ci"], pb["pi"], pb["obs"], pb["K"], start)
    ref = oracle.ba_normal_eq(pb["x"], F, P, pb["ci"], pb["pi"], pb["obs"], pb["K"], start)
    assert abs(ne["cost"] - ref["cost"]) <= 1e-12 * ref["cost"]
    for k in ("Hcc", "Hpp", "Hcp", "g", "r"):
        scale = np.abs(ref[k]).max()
        assert np.abs(ne[k] - ref[k]).max() <= 1e-10 * scale, k
    assert (ne["Hcc"][:3] == 0).all() and np.abs(ne["Hcc"][3:]).min(axis=(1, 2)).max() >= 0
    # unsorted observation order (camera blocks fall back to per-lane atomics) gives the same sums
    perm = np.random.default_rng(0).permutation(50000)
    ne2 = hip.ba_normal_eq(ctx, pb["x"], F, P, pb["ci"][perm], pb["pi"][perm], pb["obs"][perm], pb["K"], 0)
    ref2 = oracle.ba_normal_eq(pb["x"], F, P, pb["ci"][perm], pb["pi"][perm], pb["obs"][perm], pb["K"], 0)
    for k in ("Hcc", "Hpp", "g"):
        assert np.abs(ne2[k] - ref2[k]).max() <= 1e-10 * np.abs(ref2[k]).max(), k


def test_ba_device_and_edge_cases(oracle, ctx, landmark_path):
    import torch
    from visual_odometry_amd import hip
    pb = synth.ba_problem(F=4, P=100, visibility=0.6, seed=3)
    F, P = pb["F"], pb["P"]
    ref = oracle.ba_normal_eq(pb["x"], F, P, pb["ci"], pb["pi"], pb["obs"], pb["K"], 0)
    tx = torch.from_numpy(pb["x"]).cuda(); tci = torch.from_numpy(pb["ci"]).cuda(); tpi = torch.from_numpy(pb["pi"]).cuda()
    tobs = torch.from_numpy(pb["obs"].astype(np.float64)).cuda()
    ne = hip.ba_normal_eq(ctx, tx, F, P, tci, tpi, tobs, pb["K"], 0)
    ctx.sync()
    assert abs(float(ne["cost"].item()) - ref["cost"]) <= 1e-12 * ref["cost"]
    np.testing.assert_allclose(ne["g"].cpu().numpy(), ref["g"], rtol=1e-9, atol=1e-6)
    # zero rotation vector: the theta == 0 branch
    x0 = pb["x"].copy(); x0[0:3] = 0
    r, cost = hip.ba_residual(ctx, x0, F, P, pb["ci"], pb["pi"], pb["obs"], pb["K"])
    rr, rcost = oracle.ba_residual(x0, F, P, pb["ci"], pb["pi"], pb["obs"], pb["K"])
    np.testing.assert_allclose(r, rr, atol=1e-9); assert np.isfinite(r).all()
    ne0 = hip.ba_normal_eq(ctx, x0, F, P, pb["ci"], pb["pi"], pb["obs"], pb["K"], 0)
    ref0 = oracle.ba_normal_eq(x0, F, P, pb["ci"], pb["pi"], pb["obs"], pb["K"], 0)
    assert np.abs(ne0["Hcc"] - ref0["Hcc"]).max() <= 1e-10 * np.abs(ref0["Hcc"]).max()
    # empty observation set, bad indices
    r, cost = hip.ba_residual(ctx, pb["x"], F, P, np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros((0, 2)), pb["K"])
    assert r.size == 0 and cost == 0
    bad = pb["ci"].copy(); bad[5] = F
    with pytest.raises(ValueError):
        hip.ba_residual(ctx, pb["x"], F, P, bad, pb["pi"], pb["obs"], pb["K"])
    tb = torch.from_numpy(bad).cuda()
    r, _ = hip.ba_residual(ctx, tx, F, P, tb, tpi, tobs, pb["K"])
    ctx.sync()
    assert torch.isnan(r[10]).item() and torch.isfinite(r[12]).item()


def test_ba_many_windows_default_switch(oracle, ctx):
    """Four windows of the 10 x 5000 shape in one call (200 000 observations): above the library's own switch point,
    so the landmark blocks come from the gather path without any override; device buffers; twice the same bits."""
    import torch
    from visual_odometry_amd import hip
    pb = synth.ba_problem(F=10, P=5000, seed=0)
    F, P, Wn = pb["F"], pb["P"], 4
    cams = np.hstack([pb["x"][:6 * F] + 1e-3 * w for w in range(Wn)])
    x = np.hstack([cams, np.tile(pb["x"][6 * F:], Wn)])
    ci = np.hstack([pb["ci"] + F * w for w in range(Wn)])
    pi = np.hstack([pb["pi"] + P * w for w in range(Wn)])
    obs = np.tile(pb["obs"], (Wn, 1)).astype(np.float64)
    assert ci.size == 200000
    ref = oracle.ba_normal_eq(x, F * Wn, P * Wn, ci, pi, obs, pb["K"], 0)
    t = [torch.from_numpy(a).cuda() for a in (x, ci, pi, obs)]
    ne = hip.ba_normal_eq(ctx, t[0], F * Wn, P * Wn, t[1], t[2], t[3], pb["K"], 0)
    ne2 = hip.ba_normal_eq(ctx, t[0], F * Wn, P * Wn, t[1], t[2], t[3], pb["K"], 0)
    ctx.sync()
    for k in ("Hcc", "Hpp", "Hcp", "g", "r"):
        got = ne[k].cpu().numpy()
        assert np.abs(got - ref[k]).max() <= 1e-10 * np.abs(ref[k]).max(), k
    assert torch.equal(ne["Hpp"], ne2["Hpp"]) and torch.equal(ne["g"][6 * F * Wn:], ne2["g"][6 * F * Wn:])   # ordered sums


def test_score_properties_full_size(ctx):
    """Properties of the scoring rules that need no checker, at the full RANSAC configuration (3000 correspondences,
    2048 hypotheses): counts are the row sums of the masks; a fundamental matrix scaled by a power of two scores the
    same (every product scales exactly, the error is a ratio); permuting the correspondences permutes the mask columns;
    permuting the hypotheses permutes counts and rows."""
    from visual_odometry_amd import hip
    n, H = 3000, 2048
    sc = synth.two_view_scene(n=n, seed=21)
    Fh = synth.fundamental_hypotheses(sc, H, seed=21)
    c, m = hip.score_fundamental(ctx, sc["m1"], sc["m2"], Fh, 0.1)
    np.testing.assert_array_equal(c, m.sum(axis=1))
    c4, m4 = hip.score_fundamental(ctx, sc["m1"], sc["m2"], Fh * 4.0, 0.1)
    np.testing.assert_array_equal(c4, c); np.testing.assert_array_equal(m4, m)
    rng = np.random.default_rng(0)
    pp, ph = rng.permutation(n), rng.permutation(H)
    cp, mp = hip.score_fundamental(ctx, sc["m1"][pp], sc["m2"][pp], Fh[ph], 0.1)
    np.testing.assert_array_equal(cp, c[ph]); np.testing.assert_array_equal(mp, m[ph][:, pp])
    Rt = synth.pose_hypotheses(sc, H, seed=21)
    c, m = hip.score_pnp(ctx, sc["X"], sc["m2"], sc["K"], Rt, 1.0)
    np.testing.assert_array_equal(c, m.sum(axis=1))
    cp, mp = hip.score_pnp(ctx, sc["X"][pp], sc["m2"][pp], sc["K"], Rt[ph], 1.0)
    np.testing.assert_array_equal(cp, c[ph]); np.testing.assert_array_equal(mp, m[ph][:, pp])
    assert c.max() > 0.5 * n
