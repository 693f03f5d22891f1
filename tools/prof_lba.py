"""Times plba_lba_visual (the pre-init visual-only local BA) on a synthetic window; run under rocprofv3 for
the per-kernel split:   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_lba -- python3 tools/prof_lba.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as g

pkg = g.load_package()
K, Np, Nl = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (20, 6000, 1200)))
w = pkg.window.make_visual_window(K=K, Np=Np, Nl=Nl, n_fixed=2, seed=21)
p = pkg.new_problem()
run = lambda: p.lba_visual(w["T_kf_w"], w["kf_loc"], w["xyz"], w["pq"], w["po_pt"], w["po_kf"], w["uv"], w["lo_ln"], w["lo_kf"], w["l3"], w["cam"])
r = run()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); r = run(); ts.append(time.perf_counter() - t0)
print("lba_visual K=%d Np=%d Nl=%d obs=%d+%d: %d solves, %.3f ms per call (min of 5), %.3f ms per solve" %
      (K, Np, Nl, len(w["po_pt"]), len(w["lo_ln"]), r["iterations"], min(ts) * 1e3, min(ts) * 1e3 / max(r["iterations"], 1)))
p.close()
