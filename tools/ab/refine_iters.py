import os, sys, time, json
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from visual_odometry_amd import hip, synth
from tests.test_frame_gpu import _scene, K
from scipy.spatial.transform import Rotation
ctx = hip.Context(0)
img0, img1, pts, X, R, t = _scene(shape=synth.PARKING_SHAPE, n=1500, seed=3)
moved = None
p0 = hip.Pyramid(ctx, 1, 480, 640, 3).build(img0); p1 = hip.Pyramid(ctx, 1, 480, 640, 3).build(img1)
out, st, _ = hip.klt_track(ctx, p0, p1, pts)
Xd, xd = torch.from_numpy(X).cuda(), torch.from_numpy(out[0]).cuda()
rv0 = Rotation.from_matrix(R).as_rotvec() + 2e-3
res = {}
for it in (0, 1, 2, 3, 4, 10):
    for _ in range(20): hip.refine_pose(ctx, Xd, xd, K, rv0, t, max_iters=it)
    ts = []
    for _ in range(300):
        t0 = time.perf_counter(); r = hip.refine_pose(ctx, Xd, xd, K, rv0, t, max_iters=it); ts.append(time.perf_counter() - t0)
    res[it] = round(float(np.median(ts) * 1e6), 1)
print(json.dumps(res))
poses = {it: np.hstack(hip.refine_pose(ctx, Xd, xd, K, rv0, t, max_iters=it)[:2]) for it in (0, 1, 2, 3, 4)}
print(json.dumps({f"step{it}": float(np.abs(poses[it] - poses[it - 1]).max() / max(1.0, np.abs(poses[it]).max())) for it in (1, 2, 3, 4)}))
