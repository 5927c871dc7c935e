#!/usr/bin/env python3
"""Seed sweep of the device-resident bundle-adjustment solver: random window sizes (reduced system in LDS up
to 23 cameras, in global memory beyond), visibilities and dampings; every trial step against a dense numpy
Schur step on the CPU oracle's normal equations, every cost against the oracle's residual."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=20)
    a = ap.parse_args()
    from oracle import oracle
    from test_ba_solver_gpu import numpy_lm_step
    from visual_odometry_amd import hip, synth
    ctx = hip.Context(0)
    rng = np.random.default_rng(77)
    worst_step = worst_cost = worst_ratio = 0.0
    steps = 0
    for seed in range(a.seeds):
        F = int(rng.choice([2, 3, 5, 8, 10, 13, 20, 23, 24, 30]))
        P = int(rng.integers(8, 1500))
        pb = synth.ba_problem(F=F, P=P, visibility=float(rng.uniform(0.25, 1.0)), seed=seed, obs_noise=float(rng.uniform(0.1, 1.0)))
        ci, pi, K = pb["ci"], pb["pi"], pb["K"]
        obs = pb["obs"].astype(np.float64)
        start = int(np.count_nonzero(ci < int(F / 4) + 1))
        x = pb["x"] + np.concatenate([np.zeros(6 * (int(F / 4) + 1)), rng.normal(0, 0.005, 6 * F - 6 * (int(F / 4) + 1)),
                                      rng.normal(0, 0.05, 3 * P)])
        solver = hip.BaSolver(ctx, F, P, ci, pi, obs, K, start)
        for rounds in range(3):
            ne = oracle.ba_normal_eq(x, F, P, ci, pi, obs, K, start)
            cost = solver.linearize(x)
            assert abs(cost - float(ne["cost"])) <= 1e-9 * float(ne["cost"]), ("cost", seed, rounds)
            lam = float(rng.choice([1e-4, 1e-2, 1.0, 100.0]))
            want, cond = numpy_lm_step(ne, x, F, P, ci, pi, lam, with_condition=True)
            cost_new, solved = solver.try_step(lam)
            assert solved, ("not solved", seed, rounds, lam)
            got = solver.params(candidate=True)
            step = max(np.abs(want - x).max(), 1e-12)
            dev = float(np.abs(got - want).max() / step)
            # the cost the solver reports is the oracle's cost at the solver's own candidate (the cost at the numpy
            # step can differ by more than the parameters do when a landmark lands close to a camera plane)
            ref_cost = float(oracle.ba_residual(got, F, P, ci, pi, obs, K)[1])
            cdev = abs(cost_new - ref_cost) / ref_cost
            worst_step, worst_cost = max(worst_step, dev), max(worst_cost, float(cdev))
            # two float64 solvers of one system agree to about its condition number times the rounding unit
            tol = max(1e-6, 10 * np.finfo(np.float64).eps * cond)
            worst_ratio = max(worst_ratio, dev / tol)
            assert dev <= tol and cdev <= 1e-10, ("step", seed, rounds, F, P, lam, dev, cdev, cond)
            steps += 1
            x = want      # the next round starts from the reference step: the cases do not depend on the build's rounding
        solver.close()
    print(json.dumps({"op": "ba_solver", "seeds": a.seeds, "trial_steps": steps, "max_step_dev_rel": worst_step,
                      "max_step_dev_over_tolerance": worst_ratio, "max_cost_dev_rel": worst_cost}), flush=True)


if __name__ == "__main__":
    main()
