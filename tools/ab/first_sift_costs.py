"""Where the first SIFT call of a fresh process spends its time: the first three calls timed whole, run under
`rocprofv3 --hip-trace --stats` for the HIP API calls behind them."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t0 = time.perf_counter()
from visual_odometry_amd import hip, synth
t_import = time.perf_counter() - t0
t0 = time.perf_counter(); ctx = hip.Context(0); t_ctx = time.perf_counter() - t0
img = synth.frame_pair((480, 640), seed=1)
out = {"import_ms": t_import * 1e3, "ctx_ms": t_ctx * 1e3}
order = sys.argv[1] if len(sys.argv) > 1 else "sift"
if order == "corners_first":
    t0 = time.perf_counter(); hip.detect_corners(ctx, img[0], 1500, 0.01, 5, 0.04, hip.MIN_EIGEN); out["first_corners_ms"] = (time.perf_counter() - t0) * 1e3
for rep in range(3):
    t0 = time.perf_counter()
    kp, des = hip.sift_detect_compute(ctx, img[rep % 2])
    out[f"sift_call_{rep}_ms"] = (time.perf_counter() - t0) * 1e3
    out["keypoints"] = len(kp)
t0 = time.perf_counter(); hip.detect_corners(ctx, img[0], 1500, 0.01, 5, 0.04, hip.MIN_EIGEN); out["corners_after_ms"] = (time.perf_counter() - t0) * 1e3
print(json.dumps(out))
