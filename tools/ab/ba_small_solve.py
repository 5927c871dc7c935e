"""Trial-step time of a small window (F = 10, the LDS-resident kernels) against the kernels of large windows."""
import os, sys, json, time
sys.path.insert(0, "/root/repo")
import numpy as np
from visual_odometry_amd import hip, synth
ctx = hip.Context(0)
for F, P in ((10, 5000), (20, 5000)):
    pb = synth.ba_problem(F=F, P=P, visibility=1.0, seed=0, obs_noise=0.5)
    start = int(np.count_nonzero(pb["ci"] < int(F / 4) + 1))
    rec = {"F": F}
    for mode in ("lds", "global+panels", "global+blocked"):
        os.environ.pop("VO_BA_NO_LDS", None); os.environ.pop("VO_BA_SOLVE", None)
        if mode != "lds": os.environ["VO_BA_NO_LDS"] = "1"
        if mode == "global+blocked": os.environ["VO_BA_SOLVE"] = "blocked"
        s = hip.BaSolver(ctx, F, P, pb["ci"], pb["pi"], pb["obs"].astype(np.float64), pb["K"], start)
        c0 = s.linearize(pb["x"] + 1e-3)
        for _ in range(10): s.try_step(1e-3)
        ts = []
        for _ in range(100):
            t0 = time.perf_counter(); c, ok = s.try_step(1e-3); ts.append(time.perf_counter() - t0)
        rec[mode] = round(float(np.median(ts)) * 1e6, 1); rec[mode + "_cost"] = c
        s.close()
    print(json.dumps(rec))
