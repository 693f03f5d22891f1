"""ms per LM iteration of one BASELINE config with the record-based passes (lm_fused = 0) and the fused ones (lm_fused = 2), same window,
same protocol (5 iterations, gating, then timed stage-2 iterations replayed from the saved post-gating state): python tools/time_config.py 2"""
import sys, time
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w = pkg.window.make_config(cfg)
print("config", cfg, "observations", len(w["po_kf"]) + len(w["lo_kf"]))
for mode in (0, 2, 0, 2):
    g = pkg.new_problem(lm_fused=mode); g.upload_window(w)
    g.optimize(5); g.gate_outliers()
    g.save_state()
    best = 1e9
    for rep in range(6):
        g.restore_state()
        t0 = time.perf_counter(); s = g.optimize(iters); dt = time.perf_counter() - t0
        best = min(best, dt / max(s.trials, 1))
    print("lm_fused", mode, "fused ran" if g.debug_get("lm_fused")[0] else "record path", "%.4f ms per trial" % (best * 1e3), "trials", s.trials)
    g.close()
