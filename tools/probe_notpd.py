"""probe: a landmark whose damped block Hll + lambda I is numerically singular (one active observation, lambda = 1e-30)"""
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
from oracle import oracle as orc
w = pkg.window.make_window(6, 80, 20, imu=True, seed=5)
lev = np.zeros(len(w["po_pt"]), np.uint8)
first = {}
for e, pt in enumerate(w["po_pt"]):
    if pt in (0, 1, 2):
        if pt in first: lev[e] = 1
        first[pt] = e
for lam in (1e-30, 1e-12, 1e-6):
    for name, mk in (("oracle", lambda: orc.new_problem(user_lambda_init=lam)), ("record", lambda: pkg.new_problem(user_lambda_init=lam, lm_fused=0)), ("fused", lambda: pkg.new_problem(user_lambda_init=lam, lm_fused=2))):
        p = mk(); p.upload_window(w); p.set_levels(pkg.abi.EDGE_POINT, lev)
        st = p.optimize(4); tr = p.trace(); pts = p.get_points()
        print("%g %-7s iters %d trials %d fails %d stop %d chi2 %.6g -> %.6g  dec %s  moved pt0 %.3e" % (lam, name, st.iterations, st.trials, st.solver_failures, st.stop_reason, st.chi2_initial, st.chi2_final,
              "".join(str(r["accepted"]) for r in tr), np.abs(pts[0] - w["points"][0]).max()))
        p.close()
