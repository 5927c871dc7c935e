import sys, time, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np
from visual_odometry_amd import params
from visual_odometry_amd.utils import loadData
from visual_odometry_amd.vo_pipeline.continuousVO import ContinuousVO
params.use_dataset("PARKING")
for boot in ("sift",):
    data = loadData.synthetic_parking(60, 5)
    frames = list(data.frames); data.frames = iter(frames)
    vo = ContinuousVO(data, rescale_bootstrap=True, bootstrap_correspondences=boot)
    per = []
    while True:
        t0 = time.perf_counter()
        if vo.step() is None: break
        per.append(time.perf_counter() - t0)
    per = np.array(per) * 1e3
    kf = [k.idx for k in vo.keyframes]
    print(boot, "keyframe step ms:", [round(per[i], 1) for i in kf], "sum tracked", round(per[11:].sum(), 1), "sum non-key", round(sum(per[i] for i in range(11, 60) if i not in kf), 1))
