"""Time of one trial step (vo_ba_try_step) on windows shaped like the frame loop's (synth.ba_window_problem).
Usage: python tools/ab/ba_window_step.py [repeats]; VO_BA_SCHUR / VO_BA_SOLVE select earlier kernels."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import hip, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ctx = hip.Context(0)
res = {}
for F in (47, 77, 107):
    pb = synth.ba_window_problem(F=F, seed=1)
    P = pb["P"]
    start = int(np.count_nonzero(pb["ci"] < int(F / 4) + 1))
    k = np.bincount(pb["pi"][start:], minlength=P)
    s = hip.BaSolver(ctx, F, P, pb["ci"], pb["pi"], pb["obs"].astype(np.float64), pb["K"], start)
    s.linearize(pb["x"])
    for _ in range(3):
        s.try_step(1e-3)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); s.try_step(1e-3); ts.append(time.perf_counter() - t0)
    res[f"F{F}_P{P}_O{pb['ci'].size}"] = {"ms": round(float(np.median(ts)) * 1e3, 3), "block_pairs": int((k * (k + 1) // 2).sum())}
    s.close()
print(os.environ.get("VO_BA_SCHUR", "strips"), os.environ.get("VO_BA_SOLVE", ""), json.dumps(res))
