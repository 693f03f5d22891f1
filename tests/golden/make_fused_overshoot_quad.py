"""Fixture generator (container only): the rejected-trial windows of fused_overshoot_cases.py run through the QUAD-precision build of the
oracle (oracle/make_quad.py: plba_oracle.c in __float128, itself pinned to the 40-digit third implementation by tests/test_lm_trace.py)
and through the fp64 oracle -> fused_overshoot_quad.json: per case the quad run's LM trace and final keyframe states, the fp64 oracle's
trace and its distance from the quad run.  The arbiter of tests/test_fused_overshoot.py: the fused landmark passes
are held to the EXACT trajectory where fp64 solvers legitimately part.      python tests/golden/make_fused_overshoot_quad.py"""
import json, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
import numpy as np
import __graft_entry__ as g
from oracle import oracle as orc
import fused_overshoot_cases as fc

KEYS = ("iteration", "trial", "accepted", "solver_ok", "lam", "chi2_current", "chi2_trial", "scale", "rho")
pkg = g.load_package()
out = {"note": "quad-precision oracle (oracle/make_quad.py) and fp64 oracle on fused_overshoot_cases.py"}
for name, c in fc.CASES.items():
    w = fc.window(pkg, name)
    t0 = time.time()
    q = orc.new_quad_problem(user_lambda_init=c["lambda_init"]); q.upload_window(w); sq = q.optimize(c["iters"])
    o = orc.new_problem(user_lambda_init=c["lambda_init"]); o.upload_window(w); so = o.optimize(c["iters"])
    kq, ko = q.get_keyframes(), o.get_keyframes()
    d64 = {k: float(np.abs(kq[k] - ko[k]).max()) for k in kq}
    trq, tro = q.trace(), o.trace()
    out[name] = {"meta": dict(c, Ep=int(w["meta"]["Ep"]), El=int(w["meta"]["El"])),
                 "kf": {k: kq[k].tolist() for k in kq},
                 "trace": [{k: t[k] for k in KEYS} for t in trq],
                 "trace_fp64_oracle": [{k: t[k] for k in KEYS} for t in tro],
                 "chi2_final": sq.chi2_final, "chi2_final_fp64_oracle": so.chi2_final,
                 "fp64_oracle_abs_diff": d64}
    same = [(t["iteration"], t["trial"], t["accepted"], t["solver_ok"]) for t in trq] == [(t["iteration"], t["trial"], t["accepted"], t["solver_ok"]) for t in tro]
    print("%s: %d observations, %d trials (%d rejected), chi2 %.6g | fp64 oracle: same decisions %s, max |fp64 - quad| %.3e   [%.0f s]" % (
        name, w["meta"]["Ep"] + w["meta"]["El"], len(trq), sum(1 - t["accepted"] for t in trq), sq.chi2_final, same, max(d64.values()), time.time() - t0))
    q.close(); o.close()
json.dump(out, open(os.path.join(HERE, "fused_overshoot_quad.json"), "w"))
print("wrote fused_overshoot_quad.json")
