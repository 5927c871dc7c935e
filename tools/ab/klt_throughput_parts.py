"""64-frame tracker launch time against the iteration cap: what the template part (staging, derivatives, 2x2 system)
and one Gauss-Newton iteration cost at throughput."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from visual_odometry_amd import hip
S = 32
frames, pts = bench.make_streams(list(range(S)), 2)
ctx = hip.Context.from_torch(0)
f0, f1 = torch.from_numpy(frames[0].copy()).cuda(), torch.from_numpy(frames[1].copy()).cuda()
p = torch.from_numpy(pts.copy()).cuda()
out = (torch.empty_like(p), torch.empty((S, bench.N_KP), dtype=torch.uint8, device="cuda"), torch.empty((S, bench.N_KP), dtype=torch.float32, device="cuda"))
for lv in (0, 3):
    p0 = hip.Pyramid(ctx, S, bench.H, bench.W, lv).build(f0)
    p1 = hip.Pyramid(ctx, S, bench.H, bench.W, lv).build(f1)
    rec = {"frames": S, "max_level": lv}
    for it in (0, 1, 2, 3, 4, 10):
        for _ in range(20):
            hip.klt_track(ctx, p0, p1, p, out=out, max_iters=it)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            hip.klt_track(ctx, p0, p1, p, out=out, max_iters=it)
        b.record(); b.synchronize()
        rec[f"us_iters{it}"] = round(a.elapsed_time(b) * 1e3 / 200, 1)
    print(json.dumps(rec), flush=True)
