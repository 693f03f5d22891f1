"""Marginalization: device (marg_exact 0 / 1 / 2) against the oracle's dense eigen pseudo-inverse on far-landmark windows and on
kept blocks beyond 100 dims; prints the deviation of A', b', r0^T r0, the path taken and the call time."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as ge
pkg = ge.load_package()
from oracle import oracle as orc


sys.path.insert(0, "tests/golden")
import marg_cases


def far_window(far):
    return marg_cases.case_window(pkg, dict(far=far))


def run(w, label, iters=0, prior=None):
    if prior is not None:
        w = dict(w); w["prior"] = prior
    o = orc.new_problem(); o.upload_window(w)
    if iters: o.optimize(iters)
    t0 = time.perf_counter(); po = o.marginalize(0, 50); to = time.perf_counter() - t0
    o.close()
    for mode in (0, 1, 2):
        g = pkg.new_problem(marg_exact=mode); g.upload_window(w)
        if iters: g.optimize(iters)
        g.marginalize(0, 50)
        t0 = time.perf_counter(); pg = g.marginalize(0, 50); tg = time.perf_counter() - t0
        path = g.debug_get("marg_path")
        g.close()
        sc = np.abs(po["Ar"]).max()
        dA = np.abs(pg["Ar"] - po["Ar"]).max() / sc
        db = np.abs(pg["br"] - po["br"]).max() / max(np.abs(po["br"]).max(), 1.0)
        rr = abs(pg["r0"] @ pg["r0"] - po["r0"] @ po["r0"]) / max(abs(po["r0"] @ po["r0"]), 1e-300)
        print("%-28s mode %d n %3d m %3d path %d  dA' %.2e db' %.2e r0r0 %.2e | cert w %.2e lmin %.2e tau %.2e piv %.2e | %.2f ms (oracle %.0f ms)" % (
            label, mode, pg["n"], pg["m"], int(path[0]), dA, db, rr, path[1], path[2], path[3], path[4], tg * 1e3, to * 1e3), flush=True)
    return po


if __name__ == "__main__":
    for far in (1.0, 10.0, 1e2, 1e3, 1e4, 1e6, 1e7):
        run(far_window(far), "far %.0e" % far)
    w = pkg.window.make_window(12, 260, 50, imu=True, seed=21)
    po = run(w, "K12 seed21", iters=3)
    run(w, "K12 seed21 chained", iters=3, prior=po)
    for K, dt in ((11, 0.1), (12, 0.1), (15, 0.08), (21, 0.05)):
        w = pkg.window.make_window(K, 300, 60, imu=True, seed=77, kf_dt=dt, track=(K, K))
        po = run(w, "K%d full tracks" % K, iters=2)
        run(w, "K%d full tracks chained" % K, iters=2, prior=po)
