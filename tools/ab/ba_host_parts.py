"""Where a BundleAdjustment.bundle_adjustment call spends its time, on windows captured from the synthetic loop."""
import sys, time, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np
from visual_odometry_amd import params, hip
from visual_odometry_amd.utils import loadData
from visual_odometry_amd.vo_pipeline.continuousVO import ContinuousVO
from visual_odometry_amd.vo_pipeline import bundleAdjustment as bam
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
def med(f, n=15):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); out = f(); ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e3, 3), out
for M, lm, ci, pi, kp in calls[-3:]:
    ci = np.asarray(ci, np.int64); pi = np.asarray(pi, np.int64)
    F, P, O = M.shape[0], lm.shape[0], ci.size
    res = {"F": F, "P": P, "O": O, "dtypes": [str(M.dtype), str(lm.dtype), str(kp.dtype)]}
    res["whole"], _ = med(lambda: ba.bundle_adjustment(M, lm, ci, pi, kp), 7)
    res["camera_params"], cp = med(lambda: ba._camera_params(M))
    res["hstack"], x0 = med(lambda: np.hstack((cp.ravel(), np.asarray(lm, np.float64).ravel())))
    res["_fun"], r = med(lambda: ba._fun(x0, F, P, ci, pi, kp))
    res["square_sum"], d = med(lambda: (r ** 2).reshape(-1, 2).sum(axis=1))
    keep = ~(d > params.BA_DISTANCE_TH)
    res["filter"], (kp2, ci2, pi2) = med(lambda: (kp[keep], ci[keep], pi[keep]))
    res["preprocess"], _ = med(lambda: ba._preprocess_data(M, lm, ci, pi, kp))
    start = ba._obs_start(F, ci2)
    obs = np.asarray(kp2, np.float64)
    def mk():
        s = hip.BaSolver(vo.bundle_adjustment and __import__("visual_odometry_amd.vo_pipeline._runtime", fromlist=["x"]).context(), F, P, ci2, pi2, obs, ba.K, start)
        s.close()
    res["solver_create_close"], _ = med(mk)
    ctxr = __import__("visual_odometry_amd.vo_pipeline._runtime", fromlist=["x"]).context()
    res["lm"], _ = med(lambda: ba._levenberg_marquardt(x0.copy(), hip.BaSolver(ctxr, F, P, ci2, pi2, obs, ba.K, start), 50, 1e-4), 7)
    res["filtered_create_close"], _ = med(lambda: hip.BaSolver.filtered(ctxr, F, P, ci, pi, kp, ba.K, x0, params.BA_DISTANCE_TH, int(F / 4) + 1).close())
    print(json.dumps(res))
