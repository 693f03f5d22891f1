"""Where one reference-shaped BA call spends its wall time (upload, structure build, 5+10 iterations, gating, write-back)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
import torch
pkg = ge.load_package()
w = pkg.window.make_config(3)
def T(): torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    t0 = T(); p = pkg.new_problem(); t1 = T(); p.upload_window(w); t2 = T()
    s1 = p.optimize(5); t3 = T(); g = p.gate_outliers(pkg.window.CHI2_GATE); t4 = T(); s2 = p.optimize(10); t5 = T()
    r = pkg.protocol.results(p); t6 = T(); p.close(); t7 = T()
    print("create %.2f  upload %.2f  optimize(5) %.2f [lib %.2f]  gate %.2f  optimize(10) %.2f [lib %.2f]  results %.2f  close %.2f  total %.2f ms" % (
        (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, s1.ms_total, (t4-t3)*1e3, (t5-t4)*1e3, s2.ms_total, (t6-t5)*1e3, (t7-t6)*1e3, (t7-t0)*1e3))
