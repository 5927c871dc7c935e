"""GPU tests of the device-resident bundle-adjustment solver (vo_ba_create / vo_ba_linearize /
vo_ba_try_step / vo_ba_accept / vo_ba_params) against a dense numpy restatement of the same
damped Schur step built on the CPU oracle's normal equations."""
import numpy as np
import pytest

from visual_odometry_amd import synth

pytestmark = pytest.mark.gpu


def numpy_lm_step(ne, x, F, P, ci, pi, lam, with_condition=False):
    """One damped step from the block normal equations: D = H + lam*diag(H) (+1e-12 on camera
    diagonals), identity for blocks without Jacobian rows, Schur complement on the landmarks.
    with_condition: also the 2-norm condition number of the reduced system."""
    Hcc, Hpp, Hcp, g = ne["Hcc"], ne["Hpp"], ne["Hcp"], ne["g"]
    gc, gp = g[:6 * F].reshape(F, 6), g[6 * F:].reshape(P, 3)
    Dc = Hcc + lam * np.einsum("fij,ij->fij", Hcc, np.eye(6)) + 1e-12 * np.eye(6)
    Dp = Hpp + lam * np.einsum("pij,ij->pij", Hpp, np.eye(3))
    dead_p = np.abs(Hpp).sum(axis=(1, 2)) == 0
    dead_c = np.abs(Hcc).sum(axis=(1, 2)) == 0
    Dp[dead_p] = np.eye(3)
    Dc[dead_c] = np.eye(6)
    Dp_inv = np.linalg.inv(Dp)
    E = np.zeros((P, F, 6, 3))
    np.add.at(E, (pi, ci), Hcp)
    T = np.einsum("pcij,pjk->pcik", E, Dp_inv)
    Sb = -np.einsum("pcik,pdjk->cdij", T, E)
    Sb[np.arange(F), np.arange(F)] += Dc
    S = Sb.transpose(0, 2, 1, 3).reshape(6 * F, 6 * F)
    rhs = -gc + np.einsum("pcik,pk->ci", T, gp)
    dc = np.linalg.solve(S, rhs.ravel()).reshape(F, 6)
    dc[dead_c] = 0
    back = -gp.copy()
    np.subtract.at(back, pi, np.einsum("oji,oj->oi", Hcp, dc[ci]))
    dp = np.einsum("pij,pj->pi", Dp_inv, back)
    dp[dead_p] = 0
    if with_condition:
        return x + np.hstack([dc.ravel(), dp.ravel()]), float(np.linalg.cond(S))
    return x + np.hstack([dc.ravel(), dp.ravel()])


def _start(F, ci):
    return int(np.count_nonzero(ci < int(F / 4) + 1))


# 6 cameras: reduced system in LDS (factorised a camera at a time, or column by column: "columns"); 16: LDS beyond the default
# 64 KB; 22: the largest system the camera-at-a-time kernel takes; 23: the largest in LDS (column by column); 26 and more: reduced system kept in global memory
# and factorised in panels, one launch per panel (26: 8 panels; 60: the size of the end-to-end run's shorter windows;
# 101: its longer ones, a ragged last panel), or by the kernels kept for comparison: all panels in one workgroup
# ("blocked") and column by column ("columns")
@pytest.mark.parametrize("F,P,vis,solve", [(6, 150, 0.7, ""), (6, 150, 0.7, "columns"), (16, 300, 0.5, ""), (16, 300, 0.5, "columns"),
                                           (22, 250, 0.5, ""), (23, 250, 0.5, ""), (26, 200, 0.4, ""), (26, 200, 0.4, "columns"),
                                           (26, 200, 0.4, "blocked"), (60, 400, 0.3, ""), (60, 400, 0.3, "blocked"),
                                           (101, 600, 0.3, "")])
def test_lm_step_matches_dense_numpy(oracle, ctx, F, P, vis, solve, monkeypatch):
    from visual_odometry_amd import hip
    if solve:
        monkeypatch.setenv("VO_BA_SOLVE", solve)
    pb = synth.ba_problem(F=F, P=P, visibility=vis, seed=F, obs_noise=0.4)
    ci, pi, K = pb["ci"], pb["pi"], pb["K"]
    obs = pb["obs"].astype(np.float64)
    start = _start(F, ci)
    x = pb["x"]
    ne = oracle.ba_normal_eq(x, F, P, ci, pi, obs, K, start)
    solver = hip.BaSolver(ctx, F, P, ci, pi, obs, K, start)
    cost = solver.linearize(x)
    assert abs(cost - float(ne["cost"])) <= 1e-10 * float(ne["cost"])
    np.testing.assert_array_equal(solver.params(), x)
    for lam in (1e-3, 1.0):
        want = numpy_lm_step(ne, x, F, P, ci, pi, lam)
        cost_new, solved = solver.try_step(lam)
        assert solved
        got = solver.params(candidate=True)
        step = np.abs(want - x).max()
        assert np.abs(got - want).max() <= 1e-7 * max(step, 1e-12) + 1e-12
        ref_cost = float(oracle.ba_residual(want, F, P, ci, pi, obs, K)[1])
        assert abs(cost_new - ref_cost) <= 1e-8 * ref_cost
        # frozen cameras do not move
        nf = int(F / 4) + 1
        np.testing.assert_array_equal(got[:6 * nf], x[:6 * nf])
    # accepting the candidate makes it the current point; the next step starts from there
    solver.accept()
    x1 = solver.params()
    np.testing.assert_array_equal(x1, got)
    ne1 = oracle.ba_normal_eq(x1, F, P, ci, pi, obs, K, start)
    want2 = numpy_lm_step(ne1, x1, F, P, ci, pi, 1e-2)
    solver.try_step(1e-2)
    got2 = solver.params(candidate=True)
    assert np.abs(got2 - want2).max() <= 1e-7 * max(np.abs(want2 - x1).max(), 1e-12) + 1e-12
    solver.close()


# The Schur complement of a system kept in global memory: a block pair per lane (the default when every landmark's
# observations ascend in camera), a landmark per wavefront ("strips"), sums straight into global memory ("global").
# Windows shaped like the frame loop's (runs of neighbouring cameras, a few long tracks), a landmark seen twice by one
# camera, and observations in no camera order (which the pair kernel does not take: the solver falls back by itself).
@pytest.mark.parametrize("case", ["window", "window_wide", "twice", "shuffled", "small_no_lds"])
@pytest.mark.parametrize("schur", ["", "strips", "global"])
def test_schur_variants_match_dense_numpy(oracle, ctx, case, schur, monkeypatch):
    from visual_odometry_amd import hip
    if schur:
        monkeypatch.setenv("VO_BA_SCHUR", schur)
    if case == "small_no_lds":
        # a window that would live in LDS sent through the large-window kernels: one strip, a partial block of landmarks
        monkeypatch.setenv("VO_BA_NO_LDS", "1")
        pb = synth.ba_window_problem(F=9, births_every=3, per_birth=37, survive=0.9, seed=5)
    elif case == "window_wide":
        pb = synth.ba_window_problem(F=90, births_every=6, per_birth=40, survive=0.93, seed=4)
    else:
        pb = synth.ba_window_problem(F=40, births_every=5, per_birth=50, survive=0.88, seed=3)
    F, P, K = pb["F"], pb["P"], pb["K"]
    ci, pi, obs, x = pb["ci"], pb["pi"], pb["obs"].astype(np.float64), pb["x"]
    start = _start(F, ci)
    if case == "twice":
        # every tenth observation with Jacobian rows once more, a pixel away, right behind the original
        again = np.arange(start, ci.size)[::10]
        order = np.sort(np.concatenate([np.arange(ci.size), again]), kind="stable")
        ci, pi, obs = ci[order], pi[order], obs[order]
        obs[np.flatnonzero(np.diff(order) == 0) + 1] += 1.0
    if case == "shuffled":
        order = np.random.default_rng(0).permutation(ci.size)
        ci, pi, obs = ci[order], pi[order], obs[order]
        start = 0
    ne = oracle.ba_normal_eq(x, F, P, ci, pi, obs, K, start)
    solver = hip.BaSolver(ctx, F, P, ci, pi, obs, K, start)
    solver.linearize(x)
    for lam in (1e-3, 10.0):
        want = numpy_lm_step(ne, x, F, P, ci, pi, lam)
        _, solved = solver.try_step(lam)
        assert solved
        got = solver.params(candidate=True)
        assert np.abs(got - want).max() <= 1e-7 * max(np.abs(want - x).max(), 1e-12) + 1e-12
    solver.close()


def test_device_inputs_and_unseen_landmarks(oracle, ctx):
    import torch
    from visual_odometry_amd import hip
    pb = synth.ba_problem(F=5, P=80, visibility=0.6, seed=2)
    F, P, K = pb["F"], pb["P"] + 7, pb["K"]                       # 7 landmarks nobody observes
    ci, pi, obs = pb["ci"], pb["pi"], pb["obs"].astype(np.float64)
    x = np.hstack([pb["x"], np.arange(21, dtype=np.float64)])
    start = _start(F, ci)
    dev = torch.device("cuda", 0)
    a = hip.BaSolver(ctx, F, P, ci, pi, obs, K, start)
    b = hip.BaSolver(ctx, F, P, torch.from_numpy(ci).to(dev), torch.from_numpy(pi).to(dev), torch.from_numpy(obs).to(dev), K, start)
    ca, cb = a.linearize(x), b.linearize(torch.from_numpy(x).to(dev))
    assert ca == pytest.approx(cb, rel=1e-12)
    (na, sa), (nb, sb) = a.try_step(1e-3), b.try_step(1e-3)
    assert sa and sb and na == pytest.approx(nb, rel=1e-9)
    xa, xb = a.params(True), b.params(True)
    np.testing.assert_allclose(xa, xb, rtol=0, atol=1e-9)
    np.testing.assert_array_equal(xa[-21:], x[-21:])              # unseen landmarks keep their coordinates
    want = numpy_lm_step(oracle.ba_normal_eq(x, F, P, ci, pi, obs, K, start), x, F, P, ci, pi, 1e-3)
    assert np.abs(xa - want).max() <= 1e-7 * np.abs(want - x).max()
    assert na < ca


def test_parameter_vector_beyond_the_page_locked_paths(ctx):
    """A parameter vector of more than 1 MiB (one piece of the context's page-locked ring, and the bound of its result
    area) goes up in several pieces and comes back through the runtime's copy in vo_ba_linearize / vo_ba_params: the
    same cost as the residual call, the vector back unchanged."""
    from visual_odometry_amd import hip
    pb = synth.ba_problem(F=4, P=400, visibility=1.0, seed=4)
    F, K = pb["F"], pb["K"]
    P = 50000                                                      # 6F + 3P doubles = 1.2 MB; landmarks 400.. are unobserved
    x = np.hstack([pb["x"], np.random.default_rng(0).normal(size=3 * (P - pb["P"]))])
    ci, pi, obs = pb["ci"], pb["pi"], pb["obs"].astype(np.float64)
    s = hip.BaSolver(ctx, F, P, ci, pi, obs, K, _start(F, ci))
    cost = s.linearize(x)
    _, want = hip.ba_residual(ctx, x, F, P, ci, pi, obs, K)
    assert cost == pytest.approx(want, rel=1e-12)
    np.testing.assert_array_equal(s.params(), x)
    new_cost, solved = s.try_step(1e-3)
    assert solved and new_cost < cost
    moved = s.params(True)
    np.testing.assert_array_equal(moved[6 * F + 3 * pb["P"]:], x[6 * F + 3 * pb["P"]:])      # unobserved landmarks stay
    assert np.abs(moved - x).max() > 0
    s.close()


def test_empty_window_state_errors_and_bad_indices(ctx):
    from visual_odometry_amd import hip
    from visual_odometry_amd._lib import VoError
    K = synth.K_KITTI
    e = hip.BaSolver(ctx, 2, 3, np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros((0, 2)), K, 0)
    x = np.arange(21, dtype=np.float64)
    with pytest.raises(VoError):
        e.try_step(1e-3)                                           # nothing linearised yet
    assert e.linearize(x) == 0.0
    with pytest.raises(VoError):
        e.accept()                                                 # no candidate yet
    cost, solved = e.try_step(1e-3)
    assert cost == 0.0 and solved
    np.testing.assert_array_equal(e.params(True), x)
    with pytest.raises(ValueError):
        e.linearize(x[:-1])
    with pytest.raises(ValueError):
        hip.BaSolver(ctx, 2, 3, np.array([0, 2]), np.array([0, 1]), np.zeros((2, 2)), K, 0)   # camera 2 of 2
    with pytest.raises(ValueError):
        hip.BaSolver(ctx, 2, 3, np.array([0, 1]), np.array([0, 1, 2]), np.zeros((2, 2)), K, 0)


@pytest.mark.parametrize("F,P,vis,n_bad", [(6, 150, 0.7, 25), (30, 500, 0.4, 200), (12, 200, 0.6, 0), (1, 40, 1.0, 3)])
def test_filtered_solver_equals_host_prefilter(ctx, F, P, vis, n_bad):
    """vo_ba_create_filtered (residual at x0, outlier filter, compaction, landmark lists on the device) against the
    same steps on the host followed by vo_ba_create: kept count and frozen count exactly, cost and one trial step to
    the last bits."""
    from visual_odometry_amd import hip, params
    from visual_odometry_amd.vo_pipeline.bundleAdjustment import BundleAdjustment
    pb = synth.ba_problem(F=F, P=P, visibility=vis, seed=100 + F, obs_noise=0.4)
    ci, pi, K = pb["ci"], pb["pi"], pb["K"]
    obs = pb["obs"].astype(np.float64).copy()
    rng = np.random.default_rng(F)
    bad = rng.choice(ci.size, size=min(n_bad, ci.size), replace=False)
    obs[bad] += rng.uniform(15, 60, (bad.size, 2)) * rng.choice([-1, 1], (bad.size, 2))     # beyond sqrt(BA_DISTANCE_TH) = 10 px
    x0 = pb["x"] + rng.normal(0, 1e-3, pb["x"].size)
    n_frozen = int(F / 4) + 1
    # host route
    r, _ = hip.ba_residual(ctx, x0, F, P, ci, pi, obs, K)
    d = r[0::2] ** 2 + r[1::2] ** 2
    keep = ~(d > params.BA_DISTANCE_TH)
    assert keep.sum() == ci.size - bad.size or n_bad == 0 or keep.sum() < ci.size
    ci2, pi2, obs2 = ci[keep], pi[keep], obs[keep]
    start = int(np.count_nonzero(ci2 < n_frozen))
    a = hip.BaSolver(ctx, F, P, ci2, pi2, obs2, K, start)
    b = hip.BaSolver.filtered(ctx, F, P, ci, pi, obs, K, x0, params.BA_DISTANCE_TH, n_frozen)
    try:
        assert b.O == ci2.size and b.obs_start == start
        # float64 sums that pass through atomics (cost, camera blocks) differ in the last bits from run to run
        ca, cb = a.linearize(x0), b.linearize(x0)
        assert ca == pytest.approx(cb, rel=1e-12)
        if start < ci2.size:
            (na, sa), (nb, sb) = a.try_step(1e-3), b.try_step(1e-3)
            assert sa == sb and na == pytest.approx(nb, rel=1e-11)
            np.testing.assert_allclose(a.params(candidate=True), b.params(candidate=True), rtol=0, atol=1e-9)
    finally:
        a.close(); b.close()
    # float32 observations (the pipeline's keypoints) are widened on the device: same solver as widening on the host
    o32 = obs.astype(np.float32)
    a = hip.BaSolver.filtered(ctx, F, P, ci, pi, o32.astype(np.float64), K, x0, params.BA_DISTANCE_TH, n_frozen)
    b = hip.BaSolver.filtered(ctx, F, P, ci, pi, o32, K, x0, params.BA_DISTANCE_TH, n_frozen)
    try:
        assert (a.O, a.obs_start) == (b.O, b.obs_start)
        assert a.linearize(x0) == pytest.approx(b.linearize(x0), rel=1e-12)
    finally:
        a.close(); b.close()
    # device-resident inputs (float64 and float32 observations)
    import torch
    for o in (obs, o32):
        a = hip.BaSolver.filtered(ctx, F, P, ci, pi, o, K, x0, params.BA_DISTANCE_TH, n_frozen)
        b = hip.BaSolver.filtered(ctx, F, P, torch.from_numpy(ci).cuda(), torch.from_numpy(pi).cuda(), torch.from_numpy(o).cuda(), K,
                                  torch.from_numpy(x0).cuda(), params.BA_DISTANCE_TH, n_frozen)
        try:
            assert (a.O, a.obs_start) == (b.O, b.obs_start)
            assert a.linearize(x0) == pytest.approx(b.linearize(x0), rel=1e-12)
        finally:
            a.close(); b.close()
    # the drop-in class takes the device route
    if F >= 4:
        from scipy.spatial.transform import Rotation
        M = np.zeros((F, 3, 4))
        cams = x0[:6 * F].reshape(F, 6)
        M[:, :, :3] = Rotation.from_rotvec(cams[:, :3]).as_matrix()
        M[:, :, 3] = cams[:, 3:]
        adj_M, adj_L = BundleAdjustment(K).bundle_adjustment(M, x0[6 * F:].reshape(P, 3), ci, pi, obs)
        assert np.isfinite(adj_M).all() and np.isfinite(adj_L).all()
        np.testing.assert_allclose(adj_M[:n_frozen], M[:n_frozen], atol=1e-9)      # frozen cameras stay


def test_filtered_solver_edge_cases(ctx):
    from visual_odometry_amd import hip
    pb = synth.ba_problem(F=5, P=60, visibility=0.8, seed=3, obs_noise=0.3)
    ci, pi, K, x0 = pb["ci"], pb["pi"], pb["K"], pb["x"]
    obs = pb["obs"].astype(np.float64)
    far = obs + 500.0                                                   # every observation is an outlier
    s = hip.BaSolver.filtered(ctx, 5, 60, ci, pi, far, K, x0, 100.0, 2)
    assert s.O == 0 and s.obs_start == 0
    s.close()
    s = hip.BaSolver.filtered(ctx, 5, 60, ci[:0], pi[:0], obs[:0], K, x0, 100.0, 2)      # no observations at all
    assert s.O == 0
    s.close()
    r, _ = hip.ba_residual(ctx, x0, 5, 60, ci, pi, obs, K)
    kept = int(np.count_nonzero(~(r[0::2] ** 2 + r[1::2] ** 2 > 100.0)))
    s = hip.BaSolver.filtered(ctx, 5, 60, ci, pi, obs, K, x0, 100.0, 9)                  # every camera frozen
    assert s.O == kept and s.obs_start == kept
    assert s.linearize(x0) > 0
    s.close()
    bad = pi.copy(); bad[7] = 60
    with pytest.raises(ValueError):
        hip.BaSolver.filtered(ctx, 5, 60, ci, bad, obs, K, x0, 100.0, 2)
    with pytest.raises(ValueError):
        hip.BaSolver.filtered(ctx, 5, 60, ci, pi[:-1], obs, K, x0, 100.0, 2)


@pytest.mark.parametrize("F,P,vis,flat", [(6, 150, 0.7, True), (12, 300, 0.5, True), (30, 600, 0.3, True), (8, 200, 0.6, False),
                                          (45, 500, 0.25, True)])
def test_device_side_loop_equals_host_loop(ctx, F, P, vis, flat):
    """vo_ba_solve (accept / reject, damping and the stop rules decided by a device thread, trial steps queued without
    the host in between) against the same loop driven from the host over vo_ba_linearize / try_step / accept: same
    number of trial and accepted steps, same final cost and parameters (float64 sums through atomics differ in the last
    bits from run to run, so equality is to 1e-9, and a run whose decision hangs on such bits is not among the cases)."""
    from visual_odometry_amd import hip
    from visual_odometry_amd.vo_pipeline.bundleAdjustment import BundleAdjustment
    pb = synth.ba_problem(F=F, P=P, visibility=vis, seed=300 + F, obs_noise=0.4)
    ci, pi, K, obs = pb["ci"], pb["pi"], pb["K"], pb["obs"].astype(np.float64)
    front = pb["x"][6 * F:].reshape(P, 3)[pi, 2] - 0.5 * ci > 3.0        # the cameras advance 0.5 per frame: keep what lies ahead
    ci, pi, obs = ci[front], pi[front], obs[front]
    rng = np.random.default_rng(F)
    x0 = pb["x"].copy()
    x0[6 * (int(F / 4) + 1):6 * F] += rng.normal(0, 5e-3, 6 * F - 6 * (int(F / 4) + 1))
    x0[6 * F:] += rng.normal(0, 0.05, 3 * P)
    start = int(np.count_nonzero(ci < int(F / 4) + 1))

    class Counting:
        def __init__(self, s):
            self.s, self.O, self.trials, self.accepted = s, s.O, 0, 0

        def linearize(self, x):
            self.c0 = self.s.linearize(x)
            return self.c0

        def try_step(self, lam):
            self.trials += 1
            self.last = self.s.try_step(lam)
            return self.last

        def accept(self):
            self.accepted += 1
            self.cost = self.last[0]
            self.s.accept()

        def params(self):
            return self.s.params()

        def close(self):
            self.s.close()

    host = Counting(hip.BaSolver(ctx, F, P, ci, pi, obs, K, start))
    x_host = BundleAdjustment(K)._levenberg_marquardt(x0, host, 50, 1e-4, stop_on_flat_reject=flat)
    dev = hip.BaSolver(ctx, F, P, ci, pi, obs, K, start)
    try:
        x_dev, st = dev.solve(x0, 50, 1e-4, 1e-3, flat)
    finally:
        dev.close()
    assert (st["trial_steps"], st["accepted"]) == (host.trials, host.accepted), (st, host.trials, host.accepted)
    assert st["cost_start"] == pytest.approx(host.c0, rel=1e-12)
    assert st["cost_final"] == pytest.approx(host.cost, rel=1e-9)
    assert st["cost_final"] < 0.9 * st["cost_start"] and st["stop_reason"] in (1, 2, 3)
    # (a landmark seen by two or three neighbouring cameras is determined along its ray to ~1e-7 of its depth only)
    np.testing.assert_allclose(x_dev, x_host, rtol=2e-6, atol=1e-6)
    # the drop-in class: both routes end at the same place
    from scipy.spatial.transform import Rotation
    M = np.zeros((F, 3, 4))
    cams = x0[:6 * F].reshape(F, 6)
    M[:, :, :3] = Rotation.from_rotvec(cams[:, :3]).as_matrix()
    M[:, :, 3] = cams[:, 3:]
    L0 = x0[6 * F:].reshape(P, 3)
    a = BundleAdjustment(K).bundle_adjustment(M, L0, ci, pi, obs, stop_on_flat_reject=flat, device_loop=True)
    b = BundleAdjustment(K).bundle_adjustment(M, L0, ci, pi, obs, stop_on_flat_reject=flat, device_loop=False)
    np.testing.assert_allclose(a[0], b[0], atol=1e-7)
    np.testing.assert_allclose(a[1], b[1], rtol=2e-6, atol=1e-6)


def test_device_side_loop_empty_and_frozen_windows(ctx):
    from visual_odometry_amd import hip
    pb = synth.ba_problem(F=5, P=60, visibility=0.8, seed=4, obs_noise=0.3)
    ci, pi, K, obs, x0 = pb["ci"], pb["pi"], pb["K"], pb["obs"].astype(np.float64), pb["x"]
    s = hip.BaSolver(ctx, 5, 60, ci[:0], pi[:0], obs[:0], K, 0)               # no observations: the start comes back
    x, st = s.solve(x0)
    s.close()
    np.testing.assert_array_equal(x, x0)
    assert st["trial_steps"] == 0
    s = hip.BaSolver(ctx, 5, 60, ci, pi, obs, K, ci.size)                     # every observation frozen: ten rejected steps
    x, st = s.solve(x0)
    s.close()
    np.testing.assert_array_equal(x, x0)
    assert st["accepted"] == 0 and st["stop_reason"] in (2, 3)
    s = hip.BaSolver(ctx, 5, 60, ci, pi, obs, K, int(np.count_nonzero(ci < 2)))
    with pytest.raises(ValueError):
        s.solve(x0[:-1])
    x, st = s.solve(x0, max_iters=2)                                          # the iteration cap
    assert st["accepted"] <= 2
    s.close()


def test_solve_in_two_halves_and_its_state_errors(ctx):
    """vo_ba_solve_begin / _finish: the same result as the one call (to the order of the solver's float64 atomics), other
    contexts usable in between, a longer first round changing nothing but the queue; a second begin before finish, a
    finish without begin and the step-by-step calls between the two are state errors."""
    from visual_odometry_amd import hip
    from visual_odometry_amd._lib import VoError
    pb = synth.ba_window_problem(F=30, births_every=5, per_birth=60, survive=0.9, seed=8)
    F, P, K = pb["F"], pb["P"], pb["K"]
    ci, pi, obs, x0 = pb["ci"], pb["pi"], pb["obs"].astype(np.float64), pb["x"]
    start = _start(F, ci)
    s = hip.BaSolver(ctx, F, P, ci, pi, obs, K, start)
    want, want_stats = s.solve(x0)
    s.close()
    side = hip.Context.on_own_stream(0)
    for first_round in (0, 2, 9):
        s = hip.BaSolver(ctx, F, P, ci, pi, obs, K, start)
        with pytest.raises(VoError):
            s.solve_finish()
        s.solve_begin(x0, first_round=first_round)
        with pytest.raises(VoError):
            s.solve_begin(x0)
        # the step-by-step calls are refused too while the queued loop owns the solver's slots
        for call in (lambda: s.linearize(x0), lambda: s.try_step(1e-3), lambda: s.accept(), lambda: s.params()):
            with pytest.raises(VoError, match="has not been finished"):
                call()
        # another context does unrelated work on its own stream meanwhile
        r, cost = hip.ba_residual(side, x0, F, P, ci, pi, obs, K)
        assert np.isfinite(cost) and r.shape == (2 * ci.size,)
        got, stats = s.solve_finish()
        assert stats["trial_steps"] == want_stats["trial_steps"] and stats["accepted"] == want_stats["accepted"]
        assert stats["stop_reason"] == want_stats["stop_reason"]
        assert abs(stats["cost_final"] - want_stats["cost_final"]) <= 1e-9 * want_stats["cost_final"]
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-6)
        s.close()
    side.close()
