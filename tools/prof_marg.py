"""One BA + four marginalizations of the config-4 shaped window: the command behind profiles/r02_marg_kernel_stats.csv
(rocprofv3 --kernel-trace --stats -- python3 tools/prof_marg.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import torch
pkg = ge.load_package()
w = pkg.window.make_config(3)
p = pkg.new_problem(); p.upload_window(w)
pkg.protocol.local_ba(p)
for rep in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr = p.marginalize(0, 50)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("marginalize %.3f ms (prior dim %d, dropped %d)" % ((t1 - t0) * 1e3, pr["n"], pr["m"]), flush=True)
p.close()
