"""Fixture generator (container only): the overshooting IMU windows run through the QUAD-precision build of the oracle
(oracle/make_quad.py: the same source in __float128) -> overshoot_quad.json: final keyframe states and the LM trace.  The arbiter of
VERDICT r02 item 5(b): where the fp64 oracle and the HIP path end ~1e-5 apart after eight overshooting iterations, both are held against
this.      python tests/golden/make_overshoot_quad.py"""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
import numpy as np
import __graft_entry__ as g
from oracle import oracle as orc
import overshoot_cases as oc

pkg = g.load_package()
out = {"note": "quad-precision oracle (oracle/make_quad.py), user_lambda_init = %g, optimize(%d)" % (oc.LAMBDA_INIT, oc.ITERS)}
for name, spec in oc.CASES:
    w = oc.overshoot_window(pkg, **spec)
    q = orc.new_quad_problem(user_lambda_init=oc.LAMBDA_INIT); q.upload_window(w); sq = q.optimize(oc.ITERS)
    o = orc.new_problem(user_lambda_init=oc.LAMBDA_INIT); o.upload_window(w); so = o.optimize(oc.ITERS)
    kq, ko = q.get_keyframes(), o.get_keyframes()
    d64 = max(float(np.abs(kq[k] - ko[k]).max()) for k in kq)
    tr = q.trace()
    assert [(t["iteration"], t["trial"], t["accepted"]) for t in tr] == [(t["iteration"], t["trial"], t["accepted"]) for t in o.trace()], "the fp64 oracle takes other decisions"
    out[name] = {"kf": {k: kq[k].tolist() for k in kq}, "points": q.get_points().tolist(),
                 "trace": [{k: t[k] for k in ("iteration", "trial", "accepted", "solver_ok", "lam", "chi2_current", "chi2_trial", "scale", "rho")} for t in tr],
                 "chi2_final": sq.chi2_final, "fp64_oracle_max_abs_diff": d64}
    print(name, "trials", sq.trials, "chi2", sq.chi2_final, "| fp64 oracle - quad: %.3e" % d64)
    q.close(); o.close()
json.dump(out, open(os.path.join(HERE, "overshoot_quad.json"), "w"))
print("wrote overshoot_quad.json")
