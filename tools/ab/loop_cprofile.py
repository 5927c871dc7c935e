"""cProfile of the synthetic loop (host side): where the Python time of a run goes.  usage: loop_cprofile.py sift|klt"""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import run_vo
route = sys.argv[1] if len(sys.argv) > 1 else "klt"
frames = sys.argv[2] if len(sys.argv) > 2 else "60"
sys.argv = ["run_vo.py", "--frames", frames, "--bootstrap", route]
pr = cProfile.Profile()
pr.enable()
run_vo.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue())
