"""Wall time of the whole F-RANSAC call at the bootstrap's sizes (PARKING parameters) against point count and inlier share."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import hip, synth
ctx = hip.Context(0)
for n, inl in ((2000, 0.9), (6500, 0.9), (6500, 0.7), (6500, 0.5)):
    sc = synth.two_view_scene(n=n, inlier_frac=inl, noise=0.15, seed=3)
    for _ in range(3):
        out = hip.find_fundamental_ransac(ctx, sc["m1"], sc["m2"], 0.5, 0.999, 10000)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); out = hip.find_fundamental_ransac(ctx, sc["m1"], sc["m2"], 0.5, 0.999, 10000); ts.append(time.perf_counter() - t0)
    print(json.dumps({"n": n, "inlier_frac": inl, "ms": round(float(np.median(ts)) * 1e3, 3), "inliers": int(out[1].sum())}))
