import sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from visual_odometry_amd import hip, synth
ctx = hip.Context(0); dev = torch.device("cuda", 0)
pb = synth.ba_problem(F=10, P=5000, seed=0)
F, P = pb["F"], pb["P"]; O = pb["ci"].size
tx, tci, tpi = (torch.from_numpy(pb[k]).to(dev) for k in ("x", "ci", "pi"))
tobs = torch.from_numpy(pb["obs"].astype(np.float64)).to(dev)
Wn = 64
bx = torch.cat([tx[:6 * F].repeat(Wn), tx[6 * F:].repeat(Wn)])
bci = torch.cat([tci + w * F for w in range(Wn)]); bpi = torch.cat([tpi + w * P for w in range(Wn)])
bobs = tobs.repeat(Wn, 1)
f = lambda: hip.ba_normal_eq(ctx, bx, F * Wn, P * Wn, bci, bpi, bobs, pb["K"], 0)
for _ in range(5): f()
torch.cuda.synchronize()
ts = []
for _ in range(30):
    t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(json.dumps({"O": O * Wn, "median_ms": float(np.median(ts)) * 1e3}))
