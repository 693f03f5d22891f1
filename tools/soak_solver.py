"""Soak run of the solver paths against the oracle over window shapes the test-suite does not enumerate: keyframe counts that give
8..48 tiles (two- and four-chain plans, separators of every leftover width), with and without a marginalization prior, fixed
keyframes, rejections.  Prints one line per case; exits non-zero on the first disagreement."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import __graft_entry__ as g
from oracle import oracle as orc

pkg = g.load_package()
bad = 0
if len(sys.argv) > 2 and sys.argv[1] == "--random":      # python tools/soak_solver.py --random N [seed]: N random window lengths 26..260
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    rand_cases = [(int(k), 100 + int(s)) for k, s in zip(rng.integers(26, 261, int(sys.argv[2])), range(int(sys.argv[2])))]
else:
    rand_cases = None
cases = [(K, 12 * K, 3 * K, s) for K, s in rand_cases] if rand_cases else [(K, 12 * K, 3 * K, s) for K, s in ((26, 1), (30, 2), (34, 3), (38, 4), (41, 5), (47, 6), (53, 7), (58, 8), (64, 9), (71, 10), (77, 11), (85, 12), (96, 13), (110, 14), (128, 15), (150, 16), (170, 17), (215, 18))]
for K, Np, Nl, seed in cases:
    w = pkg.window.make_window(K, Np, Nl, imu=True, seed=0x50A0 + seed)
    if seed % 3 == 0:      # a prior from a previous BA of the same window (forced separators in the chain elimination)
        p0 = pkg.new_problem(); p0.upload_window(w); pkg.protocol.local_ba(p0); pr = p0.marginalize(0, 50); p0.close()
        w = pkg.window.make_window(K, Np, Nl, imu=True, seed=0x50A0 + seed); w["prior"] = pr
    if seed % 4 == 1:
        w["kf"]["fixed_pvr"] = np.zeros(K, np.uint8); w["kf"]["fixed_pvr"][:2] = 1
    a = pkg.new_problem(user_lambda_init=(1e3 if seed % 5 == 0 else 0.0)); a.upload_window(w)
    b = orc.new_problem(user_lambda_init=(1e3 if seed % 5 == 0 else 0.0)); b.upload_window(w)
    sa, sb = a.optimize(5), b.optimize(5)
    ka, kb = a.get_keyframes(), b.get_keyframes()
    dP = np.abs(ka["P"] - kb["P"]).max(); dV = np.abs(ka["V"] - kb["V"]).max()
    ok = (sa.iterations, sa.trials, sa.solver_failures) == (sb.iterations, sb.trials, sb.solver_failures) and dP < 1e-7 and dV < 1e-6 and abs(sa.chi2_final - sb.chi2_final) <= 1e-7 * abs(sb.chi2_final)
    print("K=%3d dense_dim %4d twin %d band %d launches %2d | it %d/%d trials %d/%d chi2 %.6e / %.6e dP %.1e dV %.1e %s" %
          (K, int(a.debug_get("dense_dim")[0]), int(a.debug_get("twin")[0]), int(a.debug_get("band")[0]), int(a.debug_get("fact_launches")[0]),
           sa.iterations, sb.iterations, sa.trials, sb.trials, sa.chi2_final, sb.chi2_final, dP, dV, "ok" if ok else "MISMATCH"), flush=True)
    bad += not ok
    a.close(); b.close()
sys.exit(1 if bad else 0)
