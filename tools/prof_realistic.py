"""The reference-shaped 12-keyframe window (bench.py's realistic leg) under rocprofv3: N stage-2 iterations with the record-based
(lm_fused = 0) or the fused passes (lm_fused = 2):   rocprofv3 --kernel-trace --stats -d out -o run -- python3 tools/prof_realistic.py 2"""
import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as ge
pkg = ge.load_package()
lmf = int(sys.argv[1]) if len(sys.argv) > 1 else 2
w = pkg.window.make_window(12, 2000, 400, imu=True, seed=0x5EED00C0, kf_dt=0.1, track=(6, 12), revisit=0.2)
g = pkg.new_problem(lm_fused=lmf); g.upload_window(w)
g.optimize(5); g.gate_outliers(); g.save_state()
t0 = time.perf_counter(); n = 0
for rep in range(30):
    g.restore_state(); n += g.optimize(10).trials
print("lm_fused %d: %.4f ms per trial, dense dim %d twin %d band %d" % (lmf, (time.perf_counter() - t0) / n * 1e3, g.debug_get("dense_dim")[0], g.debug_get("twin")[0], g.debug_get("band")[0]))
