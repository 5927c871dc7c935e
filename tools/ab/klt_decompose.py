"""Single-frame tracker launch time against pyramid depth and iteration cap (where the latency of one call goes)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from visual_odometry_amd import hip
frames, pts = bench.make_streams([0], 2)
ctx = hip.Context.from_torch(0)
f0, f1 = torch.from_numpy(frames[0].copy()).cuda(), torch.from_numpy(frames[1].copy()).cuda()
p = torch.from_numpy(pts.copy()).cuda()
out = (torch.empty_like(p), torch.empty((1, bench.N_KP), dtype=torch.uint8, device="cuda"), torch.empty((1, bench.N_KP), dtype=torch.float32, device="cuda"))
for variant in ("1", "3"):
    os.environ["VO_KLT_VARIANT"] = variant
    for lv in (0, 1, 3):
        p0 = hip.Pyramid(ctx, 1, bench.H, bench.W, lv).build(f0)
        p1 = hip.Pyramid(ctx, 1, bench.H, bench.W, lv).build(f1)
        rec = {"variant": variant, "max_level": lv}
        for it in (0, 1, 2, 10):
            for _ in range(10):
                hip.klt_track(ctx, p0, p1, p, out=out, max_iters=it)
            torch.cuda.synchronize()
            ts = []
            for _ in range(100):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); hip.klt_track(ctx, p0, p1, p, out=out, max_iters=it); b.record(); b.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            rec[f"us_iters{it}"] = round(float(np.median(ts)), 1)
        print(json.dumps(rec), flush=True)
