"""Wall time of plba_marginalize on the config-4 shaped window (first call pays device allocations; later calls reuse the pool)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
import torch
pkg = ge.load_package()
# `realistic`: the reference's own window (12 keyframes, tracks over 6 .. 12: 105 kept dims) instead of the configs[3] shape (60 kept dims)
w = pkg.window.make_window(12, 2000, 400, imu=True, seed=0x5EED00C0, kf_dt=0.1, track=(6, 12), revisit=0.2) if "realistic" in sys.argv else pkg.window.make_config(3)
for rep in range(4):
    p = pkg.new_problem(); p.upload_window(w)
    pkg.protocol.local_ba(p)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr = p.marginalize(0, 50)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    p0path = p.debug_get("marg_path")[0]
    stamps = [int(x) for x in p.debug_get("dbgbuf")[56:63]]; live = [int(x) for x in p.debug_get("dbgbuf")[:6]]
    p.close()
    nz = int((np.abs(pr["Ar"]).sum(axis=0) == 0).sum())
    print("marginalize %.2f ms (prior dim %d, %d exactly zero columns, path %s)" % ((t1 - t0) * 1e3, pr["n"] if "n" in pr else -1, nz, "dense" if p0path else "block-wise"), "pre-rotation cycles [tridiag, QL, rotate, chase, -, passes, rotations]", stamps, "live columns per sweep", live)
