import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
import torch
pkg = ge.load_package()
w = pkg.window.make_config(3)
def T(): torch.cuda.synchronize(); return time.perf_counter()
def e2e(tag, n=4):
    ts = []
    for _ in range(n):
        t0 = T(); p = pkg.new_problem(); p.upload_window(w); t2 = T()
        s1 = p.optimize(5); t3 = T(); g = p.gate_outliers(pkg.window.CHI2_GATE); s2 = p.optimize(10); t5 = T()
        r = pkg.protocol.results(p); p.close(); t7 = T()
        ts.append((t3 - t2) * 1e3)
    print("%-40s opt5: %s" % (tag, " ".join("%.1f" % x for x in ts)), flush=True)
e2e("A: nothing else alive")
P1 = pkg.new_problem(); P1.upload_window(w)
e2e("B1: second problem created, never run")
P1.close()
e2e("B3: closed")
P1 = pkg.new_problem()
e2e("C: second problem created, NOTHING uploaded")
P1.close()
P1 = pkg.new_problem(); P1.upload_window(w); P1.optimize(2)
e2e("D: second problem ran optimize(2)")
