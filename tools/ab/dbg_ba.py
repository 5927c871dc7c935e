import sys, ctypes, numpy as np
sys.path.insert(0, "/root/repo")
from visual_odometry_amd import hip, synth, _lib
ctx = hip.Context(0)
L = ctypes.CDLL(_lib.LIB_PATH)
pb = synth.ba_problem(F=60, P=8000, visibility=0.4, seed=1, obs_noise=0.4)
s = hip.BaSolver(ctx, 60, 8000, pb["ci"], pb["pi"], pb["obs"].astype(np.float64), pb["K"], int(np.count_nonzero(pb["ci"] < 16)))
s.linearize(pb["x"])
for _ in range(3): s.try_step(1e-3)
a = (ctypes.c_longlong * 8)(); L.vo_dbg_read(a); a0 = list(a)
for _ in range(10): s.try_step(1e-3)
L.vo_dbg_read(a)
print([ (x - y) / 10 / 100.0 for x, y in zip(list(a), a0)], "us per call: init, tri, rows, trailing, backsub")
