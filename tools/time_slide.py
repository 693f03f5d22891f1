"""A BA call on a slid window against the same window through a fresh upload (bench.py slide_leg); `--laps`: prepare()'s laps on stderr."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import __graft_entry__ as g
pkg = g.load_package()
if "--laps" in sys.argv:
    plain = pkg.new_problem
    pkg.new_problem = lambda **o: plain(diag=1, **o)
print("configs[2] shape", json.dumps(bench.slide_leg(pkg, 50, 20000, 4000)), flush=True)
print("12-keyframe window", json.dumps(bench.slide_leg(pkg, 12, 2000, 400, kf_dt=0.1, track=(6, 12), revisit=0.2)), flush=True)
