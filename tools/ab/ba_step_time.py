import os, sys, time, json
import numpy as np
sys.path.insert(0, "/root/repo")
from visual_odometry_amd import hip, synth
ctx = hip.Context(0)
res = {}
for F, P, vis in ((60, 8000, 0.4), (60, 20000, 0.25), (26, 3000, 0.5), (100, 16000, 0.09)):
    pb = synth.ba_problem(F=F, P=P, visibility=vis, seed=1, obs_noise=0.4)
    start = int(np.count_nonzero(pb["ci"] < int(F / 4) + 1))
    s = hip.BaSolver(ctx, F, P, pb["ci"], pb["pi"], pb["obs"].astype(np.float64), pb["K"], start)
    s.linearize(pb["x"])
    for _ in range(5): s.try_step(1e-3)
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); s.try_step(1e-3); ts.append(time.perf_counter() - t0)
    res[f"F{F}_P{P}_O{pb['ci'].size}"] = round(float(np.median(ts)) * 1e3, 3)
    s.close()
print(os.environ.get("VO_BA_SCHUR", "strips"), os.environ.get("VO_BA_SOLVE", ""), json.dumps(res))
