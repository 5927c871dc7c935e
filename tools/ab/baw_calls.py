"""Per-call wall time of ContinuousVO._ba_window over the 60-frame loop, with and without Python's cyclic GC."""
import gc, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd.vo_pipeline import continuousVO as cvo
import run_vo
if os.environ.get("NO_GC"):
    gc.disable()
orig = cvo.ContinuousVO._ba_window
times = []
def wrapped(self, *a, **k):
    t0 = time.perf_counter(); out = orig(self, *a, **k); times.append(round((time.perf_counter() - t0) * 1e3, 3)); return out
cvo.ContinuousVO._ba_window = wrapped
sys.argv = ["run_vo.py", "--frames", "60", "--bootstrap", "klt"]
run_vo.main()
print(json.dumps({"gc": not os.environ.get("NO_GC"), "ba_window_ms": times, "gc_counts": gc.get_count()}))
