import os, sys, time, json
import numpy as np
sys.path.insert(0, "/root/repo")
from visual_odometry_amd import hip, synth
ctx = hip.Context(0)
d0, d1, _ = synth.sift_like_descriptors(3000, 3000, seed=5)
res = {}
for name, (a, b) in {"3000": (d0, d1), "1500": (d0[:1500], d1[:1500])}.items():
    for _ in range(20): hip.knn_match2(ctx, a, b, 0.8)
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); hip.knn_match2(ctx, a, b, 0.8); ts.append(time.perf_counter() - t0)
    res[name] = [round(float(np.median(ts)) * 1e3, 3), round(float(np.mean(ts)) * 1e3, 3), round(float(np.max(ts)) * 1e3, 3)]
print(os.environ.get("VO_HIP_LIB", "")[-12:], json.dumps(res))
