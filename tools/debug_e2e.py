import sys, time, os, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
import torch
pkg = ge.load_package()
w = pkg.window.make_config(3)
def T(): torch.cuda.synchronize(); return time.perf_counter()
def e2e(tag, n=3):
    for _ in range(n):
        t0 = T(); p = pkg.new_problem(); p.upload_window(w); t2 = T()
        s1 = p.optimize(5); t3 = T(); g = p.gate_outliers(pkg.window.CHI2_GATE); s2 = p.optimize(10); t5 = T()
        r = pkg.protocol.results(p); p.close(); t7 = T()
        print("%-44s opt5 %.2f opt10 %.2f total %.2f" % (tag, (t3-t2)*1e3, (t5-t3)*1e3, (t7-t0)*1e3), flush=True)
e2e("A: nothing else alive")
P1 = pkg.new_problem(); P1.upload_window(w)
e2e("B1: second problem created, never run")
P1.optimize(1)
e2e("B2: second problem ran optimize(1)")
P1.close()
e2e("B3: second problem closed")
P1 = pkg.new_problem(); P1.upload_window(w); P1.optimize(5)
e2e("B4: second problem (new) ran optimize(5)")
time.sleep(0.5)
e2e("B5: after 0.5 s sleep")
