import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from visual_odometry_amd import hip
from visual_odometry_amd.vo_pipeline.continuousVO import window_indices
rng = np.random.default_rng(0)
trajs, kps, active, nxt = [], [], np.zeros(0, np.int32), 0
for f in range(108):
    if f % 7 == 0:
        active = np.concatenate([active[rng.random(active.size) > 0.1], np.arange(nxt, nxt + 1500, dtype=np.int32)]); nxt += 1500
    else:
        active = active[rng.random(active.size) > 0.14]
    trajs.append(active.copy()); kps.append(rng.normal(size=(active.size, 2)).astype(np.float32))
O = sum(t.size for t in trajs)
def numpy_path():
    ci, pi, top = window_indices(trajs)
    return ci, pi, np.concatenate(kps).astype(np.float32).reshape(-1, 2), top
def helper():
    return hip.ba_window_assemble(kps, trajs)
for fn in (numpy_path, helper):
    fn()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    print(fn.__name__, O, round(np.median(ts) * 1e3, 3), "ms")
