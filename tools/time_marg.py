"""Wall time of plba_marginalize on the config-4 shaped window (first call pays device allocations; later calls reuse the pool)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
import torch
pkg = ge.load_package()
w = pkg.window.make_config(3)
for rep in range(4):
    p = pkg.new_problem(); p.upload_window(w)
    pkg.protocol.local_ba(p)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr = p.marginalize(0, 50)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    p.close()
    print("marginalize %.2f ms (prior dim %d)" % ((t1 - t0) * 1e3, pr["n"] if "n" in pr else -1))
