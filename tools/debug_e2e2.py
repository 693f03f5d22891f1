import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
import torch
pkg = ge.load_package()
w = pkg.window.make_config(3)
def T(): torch.cuda.synchronize(); return time.perf_counter()
P1 = pkg.new_problem(); P1.upload_window(w); P1.optimize(1)
for variant in ("none", "gate", "opt10", "get_kf", "get_points", "get_lines", "gate+opt10+results", "none"):
    ts = []
    for rep in range(4):
        t0 = T(); p = pkg.new_problem(); p.upload_window(w); t1 = T()
        s1 = p.optimize(5); t2 = T()
        if "gate" in variant: p.gate_outliers(pkg.window.CHI2_GATE)
        if "opt10" in variant: p.optimize(10)
        if variant == "get_kf": p.get_keyframes()
        if variant == "get_points": p.get_points()
        if variant == "get_lines": p.get_lines()
        if "results" in variant: pkg.protocol.results(p)
        p.close()
        ts.append((t2 - t1) * 1e3)
    print("%-22s optimize(5) incl. prepare per rep: %s" % (variant, " ".join("%.2f" % x for x in ts)), flush=True)
