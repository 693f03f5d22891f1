"""Fused landmark-major passes (lm_fused = 1) against the record-based passes (lm_fused = 0) and the oracle: the built system, one
solve, then the LM protocol."""
import sys
import numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as ge
pkg = ge.load_package()
from oracle import oracle as orc


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)) if b.size else 0.0


def check(w, label, lam=5.0):
    g1 = pkg.new_problem(lm_fused=2); g1.upload_window(w)
    g0 = pkg.new_problem(lm_fused=0); g0.upload_window(w)
    o = orc.new_problem(); o.upload_window(w)
    for q in (g1, g0, o): q.debug_build(lam, True)
    print(label, "fused:", g1.debug_get("lm_fused"), "| record path:", g0.debug_get("lm_fused"))
    for name in ("chi2", "maxdiag", "err_pt", "err_ln", "hll_pt", "bl_pt", "hll_ln", "bl_ln", "bp", "bschur", "Hschur", "x"):
        a1, a0, b = g1.debug_get(name), g0.debug_get(name), o.debug_get(name)
        print("   %-8s fused vs oracle %.2e   record vs oracle %.2e   fused vs record %.2e" % (name, rel(a1, b), rel(a0, b), rel(a1, a0)), flush=True)
    for q in (g1, g0, o): q.close()
    res = {}
    for name, mk in (("fused", lambda: pkg.new_problem(lm_fused=2)), ("record", lambda: pkg.new_problem(lm_fused=0)), ("oracle", orc.new_problem)):
        q = mk(); q.upload_window(w)
        r = pkg.protocol.local_ba(q)
        res[name] = (r, pkg.protocol.results(q), q.trace()); q.close()
    for name in ("fused", "record"):
        r, out, tr = res[name]; ro, oo, tro = res["oracle"]
        print("   protocol %-6s: gated %s vs %s  trials %d+%d vs %d+%d  chi2 %.9e vs %.9e  dP %.2e dV %.2e dq %.2e dpts %.2e accept-seq equal %s" % (
            name, r["gated"], ro["gated"], r["stage1"].trials, r["stage2"].trials, ro["stage1"].trials, ro["stage2"].trials, r["stage2"].chi2_final, ro["stage2"].chi2_final,
            np.abs(out["P"] - oo["P"]).max(), np.abs(out["V"] - oo["V"]).max(), np.abs(out["q"] - oo["q"]).max(), np.abs(out["points"] - oo["points"]).max(),
            [t["accepted"] for t in tr] == [t["accepted"] for t in tro]), flush=True)


if __name__ == "__main__":
    check(pkg.window.make_window(6, 80, 20, imu=True, seed=5), "K6 small")
    check(pkg.window.make_window(12, 300, 60, imu=True, seed=0x5EED00AA), "K12")
    check(pkg.window.make_config(3, scale=0.1), "config3 x0.1")
    check(pkg.window.make_config(2, scale=0.1), "config2 x0.1 (no IMU)")
