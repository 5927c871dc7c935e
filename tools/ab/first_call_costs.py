"""First-call against steady-state time of the bundle-adjustment entry points in a fresh process (one-time costs:
device code loading, buffer growth)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import hip, synth
t0 = time.perf_counter(); ctx = hip.Context(0); t_ctx = time.perf_counter() - t0
img = synth.frame_pair((480, 640), seed=1)
t0 = time.perf_counter(); hip.detect_corners(ctx, img[0], 1500, 0.01, 5, 0.04, hip.MIN_EIGEN); t_corners = time.perf_counter() - t0
out = {"ctx_ms": t_ctx * 1e3, "first_corners_ms": t_corners * 1e3}
for rep in range(3):
    pb = synth.ba_problem(F=40, P=6000, visibility=0.2, seed=rep, obs_noise=0.4)
    rec = {}
    t0 = time.perf_counter()
    s = hip.BaSolver.filtered(ctx, 40, 6000, pb["ci"], pb["pi"], pb["obs"], pb["K"], pb["x"], 100.0, 11)
    rec["create"] = time.perf_counter() - t0
    t0 = time.perf_counter(); s.linearize(pb["x"]); rec["linearize"] = time.perf_counter() - t0
    t0 = time.perf_counter(); s.try_step(1e-3); rec["try_step"] = time.perf_counter() - t0
    t0 = time.perf_counter(); s.accept(); rec["accept"] = time.perf_counter() - t0
    t0 = time.perf_counter(); s.params(); rec["params"] = time.perf_counter() - t0
    t0 = time.perf_counter(); s.close(); rec["close"] = time.perf_counter() - t0
    out[f"ba_call_{rep}"] = {k: round(v * 1e3, 3) for k, v in rec.items()}
print(json.dumps(out))
