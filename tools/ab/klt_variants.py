"""Tracker launch time by kernel variant and batch size (KITTI-shaped frames, 2000 keypoints each, device-resident)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from visual_odometry_amd import hip
frames, pts = bench.make_streams(list(range(8)), 2)
ctx = hip.Context.from_torch(0)
for B in (1, 2, 3, 4, 6, 8, 16):
    rep = (B + 7) // 8
    f0 = torch.from_numpy(np.tile(frames[0], (rep, 1, 1))[:B].copy()).cuda()
    f1 = torch.from_numpy(np.tile(frames[1], (rep, 1, 1))[:B].copy()).cuda()
    p = torch.from_numpy(np.tile(pts, (rep, 1, 1))[:B].copy()).cuda()
    p0 = hip.Pyramid(ctx, B, bench.H, bench.W, bench.MAX_LEVEL).build(f0)
    p1 = hip.Pyramid(ctx, B, bench.H, bench.W, bench.MAX_LEVEL).build(f1)
    out = (torch.empty_like(p), torch.empty((B, bench.N_KP), dtype=torch.uint8, device="cuda"),
           torch.empty((B, bench.N_KP), dtype=torch.float32, device="cuda"))
    rec = {"frames": B, "keypoints": B * bench.N_KP}
    for variant in ("1", "3"):
        os.environ["VO_KLT_VARIANT"] = variant
        for _ in range(20):
            hip.klt_track(ctx, p0, p1, p, out=out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(100):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); hip.klt_track(ctx, p0, p1, p, out=out); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        rec["us_variant" + variant] = round(float(np.median(ts)), 1)
    print(json.dumps(rec), flush=True)
