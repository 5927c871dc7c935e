"""GPU minimal solvers (one sample per lane) against the independent numpy solvers
and planted geometry.  Solutions are compared as sets: same count, each matched
within solver precision (both are float64; algorithms differ)."""
import numpy as np
import pytest

from visual_odometry_amd import synth

pytestmark = pytest.mark.gpu


def _subsets(n, k, S, seed):
    rng = np.random.default_rng(seed)
    return np.stack([rng.choice(n, k, replace=False) for _ in range(S)]).astype(np.int32)


def test_seven_point_matches_numpy(ctx):
    from oracle import solvers_np
    from visual_odometry_amd import hip
    sc = synth.two_view_scene(n=500, noise=0.3, inlier_frac=0.7, seed=3)
    sub = _subsets(500, 7, 300, 1)
    F, ns = hip.seven_point_solve(ctx, sc["m1"], sc["m2"], sub)
    assert F.shape == (300, 3, 3, 3) and ns.shape == (300,)
    ref, owner = solvers_np.seven_point(sc["m1"][sub].astype(np.float64), sc["m2"][sub].astype(np.float64))
    nref = np.bincount(owner, minlength=300)
    assert (ns == nref).mean() > 0.98           # root multiplicity may differ for near-double roots
    m1h = np.hstack([sc["m1"], np.ones((500, 1))]).astype(np.float64)
    m2h = np.hstack([sc["m2"], np.ones((500, 1))]).astype(np.float64)
    matched = total = 0
    for s in range(300):
        assert (F[s, ns[s]:] == 0).all()
        for k in range(ns[s]):
            Fg = F[s, k]
            # every GPU solution is rank deficient and satisfies the 7 epipolar constraints
            res = np.abs(np.einsum("ni,ij,nj->n", m2h[sub[s]], Fg, m1h[sub[s]])).max()
            assert res < 1e-6 * np.abs(Fg).max() * 1e6 * 1e-3
            sv = np.linalg.svd(Fg, compute_uv=False)
            assert sv[2] < 1e-7 * sv[0]
            total += 1
            cand = ref[owner == s]
            if cand.shape[0] and min(np.abs(c / np.linalg.norm(c) - Fg / np.linalg.norm(Fg)).max() for c in cand) < 1e-6:
                matched += 1
    assert matched >= 0.98 * total


def test_seven_point_planted_exact(ctx):
    from visual_odometry_amd import hip
    sc = synth.two_view_scene(n=70, noise=0.0, inlier_frac=1.0, seed=11)
    sub = np.arange(70, dtype=np.int32).reshape(10, 7)
    F, ns = hip.seven_point_solve(ctx, sc["m1"], sc["m2"], sub)
    F0 = sc["F"] / np.linalg.norm(sc["F"])
    for s in range(10):
        assert ns[s] >= 1
        d = min(min(np.abs(F[s, k] / np.linalg.norm(F[s, k]) - F0).max(), np.abs(F[s, k] / np.linalg.norm(F[s, k]) + F0).max())
                for k in range(ns[s]))
        assert d < 5e-3      # float32 pixel rounding of the planted correspondences


def test_p3p_matches_numpy_and_planted(ctx):
    from oracle import solvers_np
    from visual_odometry_amd import hip
    sc = synth.two_view_scene(n=400, K=synth.K_KITTI, shape=synth.KITTI_SHAPE, noise=0.0, inlier_frac=1.0, seed=5)
    sub = _subsets(400, 4, 256, 2)
    Rt, valid = hip.p3p_solve(ctx, sc["X"], sc["m2"], sc["K"], sub)
    assert Rt.shape == (256, 3, 4) and valid.dtype == np.uint8
    assert valid.mean() > 0.95
    Kinv = np.linalg.inv(sc["K"])
    rays = (Kinv @ np.hstack([sc["m2"].astype(np.float64), np.ones((400, 1))]).T).T
    rays /= np.linalg.norm(rays, axis=1, keepdims=True)
    X = sc["X"].astype(np.float64)
    good = 0
    for s in range(256):
        if not valid[s]:
            assert (Rt[s] == 0).all()
            continue
        R, t = Rt[s][:, :3], Rt[s][:, 3]
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-9)
        assert np.linalg.det(R) > 0.999
        # the chosen solution is one of the numpy solver's
        sols = solvers_np.p3p(X[sub[s][:3]], rays[sub[s][:3]])
        assert sols
        assert min(np.abs(R - R2).max() + np.abs(t - t2).max() for R2, t2 in sols) < 1e-6
        if np.abs(R - sc["R"]).max() < 1e-3 and np.abs(t - sc["t"]).max() < 2e-2:
            good += 1
    assert good > 0.9 * valid.sum()          # noise-free data: the disambiguated solution is the planted pose


def test_solver_device_buffers_and_degenerate(ctx):
    import torch
    from visual_odometry_amd import hip
    sc = synth.two_view_scene(n=100, noise=0.2, seed=8)
    sub = _subsets(100, 7, 64, 3)
    F, ns = hip.seven_point_solve(ctx, sc["m1"], sc["m2"], sub)
    Fd, nd = hip.seven_point_solve(ctx, torch.from_numpy(sc["m1"]).cuda(), torch.from_numpy(sc["m2"]).cuda(),
                                   torch.from_numpy(sub).cuda())
    ctx.sync()
    np.testing.assert_array_equal(nd.cpu().numpy(), ns); np.testing.assert_array_equal(Fd.cpu().numpy(), F)
    # degenerate samples: repeated points, out-of-range indices -> no solution, no fault
    bad = np.zeros((4, 7), np.int32); bad[1] = 5; bad[2] = [0, 1, 2, 3, 4, 5, 1000]; bad[3] = -3
    F, ns = hip.seven_point_solve(ctx, sc["m1"], sc["m2"], bad)
    assert (ns == 0).all() and (F == 0).all()
    bad4 = np.zeros((3, 4), np.int32); bad4[1] = [1, 1, 1, 2]; bad4[2] = [0, 1, 2, 999]
    Rt, valid = hip.p3p_solve(ctx, sc["X"], sc["m2"], sc["K"], bad4)
    assert (valid == 0).all() and (Rt == 0).all()
    Rt, valid = hip.p3p_solve(ctx, sc["X"], sc["m2"], sc["K"], np.zeros((0, 4), np.int32))
    assert Rt.shape == (0, 3, 4)
