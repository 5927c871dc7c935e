"""Repeated BaSolver.filtered create/close on the last window of the 120-frame synthetic run (for an API trace)."""
import sys, time, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np
from visual_odometry_amd import params, hip
from visual_odometry_amd.utils import loadData
from visual_odometry_amd.vo_pipeline.continuousVO import ContinuousVO
from visual_odometry_amd.vo_pipeline import bundleAdjustment as bam
from visual_odometry_amd.vo_pipeline import _runtime
params.use_dataset("PARKING")
data = loadData.synthetic_parking(120, 5)
vo = ContinuousVO(data, rescale_bootstrap=True, bootstrap_correspondences="klt")
calls = []
orig = bam.BundleAdjustment.bundle_adjustment
def rec(self, M, lm, ci, pi, kp, **kw):
    calls.append((np.array(M), np.array(lm), np.array(ci), np.array(pi), np.array(kp)))
    return orig(self, M, lm, ci, pi, kp, **kw)
bam.BundleAdjustment.bundle_adjustment = rec
while vo.step() is not None: pass
bam.BundleAdjustment.bundle_adjustment = orig
ba = vo.bundle_adjustment
M, lm, ci, pi, kp = calls[-1]
ci = np.asarray(ci, np.int64); pi = np.asarray(pi, np.int64)
F, P = M.shape[0], lm.shape[0]
x0 = np.hstack((ba._camera_params(M).ravel(), np.asarray(lm, np.float64).ravel()))
ctx = _runtime.context()
print("MARK start", flush=True)
ts = []
for _ in range(20):
    t0 = time.perf_counter()
    s = hip.BaSolver.filtered(ctx, F, P, ci, pi, kp, ba.K, x0, params.BA_DISTANCE_TH, int(F / 4) + 1)
    t1 = time.perf_counter()
    s.close()
    ts.append((t1 - t0, time.perf_counter() - t1))
print(json.dumps({"F": F, "P": P, "O": int(ci.size), "create_ms": round(float(np.median([a for a, _ in ts])) * 1e3, 3),
                  "close_ms": round(float(np.median([b for _, b in ts])) * 1e3, 3)}))
