"""One BA call through Boundary 1 (tools/localba_harness.cpp `time`), lap by lap: configs[2] and the reference-shaped 12-keyframe window."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import __graft_entry__ as g
pkg = g.load_package()
for name, w in (("configs[2]", pkg.window.make_config(3)),
                ("12-keyframe window", pkg.window.make_window(12, 2000, 400, imu=True, seed=0x5EED00C0, kf_dt=0.1, track=(6, 12), revisit=0.2))):
    print(name, json.dumps(bench.facade_leg(w, reps=6)), flush=True)
