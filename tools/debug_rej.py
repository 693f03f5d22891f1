import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as ge
from oracle import oracle as orc
import test_gpu_parity as t
pkg = ge.load_package()
for seed in (77, 81):
    w = t._overshoot_window(pkg, seed, 0.6, 20.0, 0.1)
    o = orc.new_problem(user_lambda_init=1e4); o.upload_window(w); o.optimize(8); to = o.trace(); o.close()
    for opts in (dict(chain_elim=1), dict(chain_elim=0), dict(chain_elim=1, profile=2), dict(chain_elim=0, profile=2)):
        g = pkg.new_problem(user_lambda_init=1e4, **opts); g.upload_window(w); g.optimize(8); tg = g.trace(); g.close()
        print("seed", seed, opts)
        for a, b in zip(tg, to):
            flag = "" if (a["accepted"] == b["accepted"]) else "  <<<< decision differs"
            print("  it %d tr %d acc %d/%d ok %d/%d lam %.6e / %.6e  cur %.9e / %.9e  trial %.9e / %.9e rel %.1e%s" % (a["iteration"], a["trial"], a["accepted"], b["accepted"], a["solver_ok"], b["solver_ok"], a["lam"], b["lam"], a["chi2_current"], b["chi2_current"], a["chi2_trial"], b["chi2_trial"], abs(a["chi2_trial"] - b["chi2_trial"]) / max(abs(b["chi2_trial"]), 1e-300), flag))
