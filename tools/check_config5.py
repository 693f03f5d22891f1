"""BASELINE configs[4] (200 KF / 200k points / 40k lines + IMU) at full size on ONE GPU: per-phase times, properties."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
import torch
pkg = ge.load_package()
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
t = time.time(); w = pkg.window.make_config(5, scale=scale); print("gen %.1fs" % (time.time() - t), w["meta"], flush=True)
for prof in (2, 0):
    g = pkg.new_problem(profile=prof)
    t = time.time(); g.upload_window(w); print("upload %.2fs" % (time.time() - t), flush=True)
    t = time.time(); s1 = g.optimize(5); torch.cuda.synchronize(); t1 = time.time() - t
    print("profile=%d stage1: %d iters %d trials %.2f ms/iter (lib %.1f ms) chi %.1f -> %.1f fails %d" % (prof, s1.iterations, s1.trials, t1 / max(s1.iterations, 1) * 1e3, s1.ms_total, s1.chi2_initial, s1.chi2_final, s1.solver_failures), flush=True)
    if prof:
        print("   phases ms/iter", dict(zip(["lin", "fact", "schur", "dense", "backsub", "trial", "xchg", "hll+red"], np.round(np.array(list(s1.ms_phase)) / max(s1.iterations, 1), 3))))
    gated = g.gate_outliers(pkg.window.CHI2_GATE)
    g.save_state()
    t = time.time(); s2 = g.optimize(10); torch.cuda.synchronize(); t2 = time.time() - t
    print("profile=%d stage2: %d iters %d trials %.2f ms/iter chi %.1f -> %.1f fails %d gated %s dense_dim %d pose_dim %d" % (prof, s2.iterations, s2.trials, t2 / max(s2.iterations, 1) * 1e3, s2.chi2_initial, s2.chi2_final, s2.solver_failures, gated, g.debug_get("dense_dim")[0], g.debug_get("pose_dim")[0]), flush=True)
    if prof:
        print("   phases ms/iter", dict(zip(["lin", "fact", "schur", "dense", "backsub", "trial", "xchg", "hll+red"], np.round(np.array(list(s2.ms_phase)) / max(s2.iterations, 1), 3))))
    g.close()
