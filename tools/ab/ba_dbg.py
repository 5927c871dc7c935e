import os, sys, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np, torch
from visual_odometry_amd import hip, synth
ctx = hip.Context.from_torch(0)
pb = synth.ba_problem(F=10, P=5000, visibility=1.0, seed=0, obs_noise=0.5)
F, P, O = 10, 5000, pb["ci"].size
start = int(np.count_nonzero(pb["ci"] < 3))
tx, tci, tpi = (torch.from_numpy(pb[k]).cuda() for k in ("x", "ci", "pi"))
tobs = torch.from_numpy(pb["obs"].astype(np.float64)).cuda()
res = {}
for dbg in (0, 8, 4, 0, 8, 4):
    os.environ["VO_BA_DBG"] = str(dbg)
    for _ in range(20): hip.ba_normal_eq(ctx, tx, F, P, tci, tpi, tobs, pb["K"], start)
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); hip.ba_normal_eq(ctx, tx, F, P, tci, tpi, tobs, pb["K"], start); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    res[str(dbg) + ("b" if str(dbg) in res else "")] = round(float(np.median(ts)), 1)
print(json.dumps(res))
