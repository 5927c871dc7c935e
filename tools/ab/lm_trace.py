"""Trial-step log of every bundle-adjustment window of the synthetic 60-frame loop: (damping, cost before, cost after, accepted)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import hip
import run_vo

log = []
orig_try, orig_lin = hip.BaSolver.try_step, hip.BaSolver.linearize
def lin(self, x):
    c = orig_lin(self, x); log.append(["lin", self.F, self.O, c]); return c
def tr(self, lam):
    c, ok = orig_try(self, lam); log.append(["try", lam, c, bool(ok)]); return c, ok
hip.BaSolver.linearize, hip.BaSolver.try_step = lin, tr
sys.argv = ["run_vo.py", "--frames", "60", "--bootstrap", "klt"]
run_vo.main()
win = []
for e in log:
    if e[0] == "lin": win.append({"F": e[1], "O": e[2], "cost0": e[3], "tries": []})
    else: win[-1]["tries"].append((e[1], e[2], e[3]))
for w in win:
    cost = w["cost0"]; seq = []
    for lam, c, ok in w["tries"]:
        acc = ok and np.isfinite(c) and c < cost
        seq.append(("A" if acc else "r") + f"{(cost - c) / cost:+.1e}")
        if acc: cost = c
    print(json.dumps({"F": w["F"], "O": w["O"], "n": len(w["tries"]), "seq": " ".join(seq)}))
