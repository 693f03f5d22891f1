"""One BA + four marginalizations of the config-4 shaped window: the command behind profiles/r02_marg_kernel_stats.csv
(rocprofv3 --kernel-trace --stats -- python3 tools/prof_marg.py)."""
import os, sys, time
import numpy as np
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
print("live columns per sweep:", [int(x) for x in p.debug_get("dbgbuf")[:32]], "noise2 %.3e" % p.debug_get("dbgbuf")[40])
print("round stamps (cycles):", [int(x) for x in p.debug_get("dbgbuf")[48:56]], "pre-rotation: tridiagonalisation, QL, V^T A V (cycles):", [int(x) for x in p.debug_get("dbgbuf")[56:63]])
w=np.linalg.eigvalsh(pr["Ar"]); print("eig(A') min %.3e max %.3e, below 1e-8: %d, |A|_F %.3e" % (w.min(), w.max(), (w<=1e-8).sum(), np.linalg.norm(pr["Ar"])))
print(np.sort(w)[:12])
p.close()
