"""Which keypoints of the benchmark's workload end with status 0, and what the CPU checker says about them."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from oracle import oracle
from visual_odometry_amd import hip
S, R = 64, 3
frames, pts = bench.make_streams(list(range(S)), R)
ctx = hip.Context.from_torch(0)
dev = [torch.from_numpy(frames[r].copy()).cuda() for r in range(R)]
pyr = [hip.Pyramid(ctx, S, bench.H, bench.W, bench.MAX_LEVEL).build(dev[r]) for r in range(R)]
dp = torch.from_numpy(pts.copy()).cuda()
for r in range(R):
    for variant in ("3", "1"):
        os.environ["VO_KLT_VARIANT"] = variant
        out, st, err = hip.klt_track(ctx, pyr[r], pyr[(r + 1) % R], dp)
        ctx.sync()
        out, st = out.cpu().numpy(), st.cpu().numpy()
        lost = np.argwhere(st == 0)
        rec = {"pair": r, "variant": variant, "lost": lost.tolist()}
        for s, k in lost:
            ro, rs, _ = oracle.klt(frames[r, s], frames[(r + 1) % R, s], pts[s], max_level=bench.MAX_LEVEL)
            rec.setdefault("oracle_status", []).append(int(rs.ravel()[k]))
            rec.setdefault("point", []).append(pts[s, k].tolist())
            rec.setdefault("oracle_out", []).append(ro[k].tolist())
            rec.setdefault("gpu_out", []).append(out[s, k].tolist())
            agree = float((st[s] == rs.ravel()).mean())
            both = (st[s] == 1) & (rs.ravel() == 1)
            rec.setdefault("stream_agree", []).append(agree)
            rec.setdefault("stream_maxdev", []).append(float(np.abs(out[s][both] - ro[both]).max()))
        print(json.dumps(rec), flush=True)
