"""Where a tracked frame's time goes: the library call itself against the Python around it, per layer.
usage: tracked_frame_parts.py [klt|sift] [frames]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import hip, params, synth                        # noqa: E402
from visual_odometry_amd.utils import loadData                            # noqa: E402
from visual_odometry_amd.vo_pipeline import continuousVO as cvo           # noqa: E402
from visual_odometry_amd.vo_pipeline.continuousVO import ContinuousVO     # noqa: E402
from visual_odometry_amd.vo_pipeline.poseEstimation import PoseEstimation # noqa: E402

route = sys.argv[1] if len(sys.argv) > 1 else "klt"
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 60
params.use_dataset("PARKING")
scene = synth.BoxScene(n_frames=n_frames, seed=5)
frames = list(scene.frames())
data = loadData.Dataset(np.asarray(scene.K, np.float32), iter(frames), np.asarray(scene.poses, np.float32))
vo = ContinuousVO(data, rescale_bootstrap=True, bootstrap_correspondences=route)
spans = {}


def timed(label, fn):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            spans.setdefault(label, []).append(time.perf_counter() - t0)
    return wrapper


lib = hip.default_context(0).lib
lib.vo_track_pnp = timed("C vo_track_pnp", lib.vo_track_pnp)
hip.track_pnp = timed("hip.track_pnp", hip.track_pnp)
PoseEstimation.track_and_pose = timed("track_and_pose", PoseEstimation.track_and_pose)
vo._process_frame = timed("_process_frame", vo._process_frame)
store = vo.keypoint_trajectories
store.tracked_to_many = timed("tracked_to_many", store.tracked_to_many)
cvo.baseline_uncertainty = timed("baseline_uncertainty", cvo.baseline_uncertainty)
cvo._tracked_frame_rows = timed("_tracked_frame_rows", cvo._tracked_frame_rows)
import gc
gc.collect(); gc.freeze()
steps = []
while True:
    t0 = time.perf_counter()
    if vo.step() is None:
        break
    steps.append(time.perf_counter() - t0)
keys = {k.idx for k in vo.keyframes}
tracked = [i for i in range(vo.frames_to_skip + 1, len(steps)) if i not in keys]
# calls of tracked (non-key) frames only: _process_frame is called once per frame from frames_to_skip + 1 on
first = vo.frames_to_skip + 1
sel = [i - first for i in tracked]
out = {"route": route, "frames": len(steps), "tracked_frames": len(tracked),
       "step_median_ms": round(float(np.median([steps[i] for i in tracked])) * 1e3, 4)}
for label, v in spans.items():
    v = np.asarray(v)
    if v.size == len(steps) - first:
        v = v[sel]
    out[label + " median ms"] = round(float(np.median(v)) * 1e3, 4)
print(json.dumps(out))
