"""Soak run of the fused landmark-major passes (lm_fused = 2) against the oracle over random window shapes the test-suite does not
enumerate: keyframe counts 3..70, landmark counts from a handful to thousands, track lengths 2..8 — and, in a third of the cases, up to 16
(wide groups) — in every mix, points only / lines
only, with and without IMU edges, marginalization priors, fixed keyframes and fixed landmarks, gating between two stages, huge and
tiny initial damping (rejections).  One line per case; exits non-zero on the first disagreement.
    python tools/soak_fused.py [N] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import __graft_entry__ as g
from oracle import oracle as orc

pkg = g.load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
FUSED = int(os.environ.get("SOAK_LM_FUSED", "2"))      # 0: the same cases on the record-based passes (to tell conditioning from a defect of the fused ones)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
for case in range(N):
    K = int(rng.choice([3, 4, 5, 6, 8, 9, 12, 16, 17, 25, 33, 40, 51, 70]))
    dens = float(rng.choice([0.5, 2, 8, 30]))
    Np = int(max(0, rng.integers(0, 2) * 0 + K * dens * rng.uniform(0.5, 2)))
    Nl = int(max(0, K * dens * rng.uniform(0.0, 0.6)))
    mode = int(rng.integers(0, 6))
    if mode == 0: Nl = 0
    if mode == 1: Np = 0; Nl = max(Nl, 4)
    if Np + Nl < 4: Np = 6
    imu = bool(rng.integers(0, 4) != 0)
    if not imu and Np + Nl < 4 * K:      # without IMU edges a window needs landmarks to be determined at all: (under-determined ones fit to chi2 = 1e-25
        Np = max(Np, 4 * K)              # and their LM decisions are rounding noise, on every path)
    lo = int(rng.integers(2, 6)); hi = int(rng.integers(lo, 9))
    if rng.integers(0, 3) == 0: lo = int(rng.integers(2, 11)); hi = int(rng.integers(max(lo, 9), 17))      # tracks over 9 .. 16 keyframes: wide groups next to standard ones
    seed = 0xF05E + case
    w = pkg.window.make_window(K, Np, Nl, imu=imu, seed=seed, track=(lo, min(hi, K)))
    tag = []
    if imu and K >= 5 and rng.integers(0, 3) == 0:      # a prior from a previous BA of the same window
        p0 = pkg.new_problem(); p0.upload_window(w); pkg.protocol.local_ba(p0); pr = p0.marginalize(0, 50); p0.close()
        w = pkg.window.make_window(K, Np, Nl, imu=imu, seed=seed, track=(lo, min(hi, K))); w["prior"] = pr; tag.append("prior")
    if rng.integers(0, 3) == 0 and K >= 4:
        w["kf"]["fixed_pvr"] = np.zeros(K, np.uint8); w["kf"]["fixed_pvr"][:int(rng.integers(1, 3))] = 1; tag.append("fixedkf")
    if rng.integers(0, 3) == 0 and Np:
        w["point_fixed"] = (rng.random(Np) < 0.15).astype(np.uint8); tag.append("fixedpt")
    if rng.integers(0, 4) == 0 and Nl:
        w["line_fixed"] = (rng.random(Nl) < 0.2).astype(np.uint8); tag.append("fixedln")
    lam = float(rng.choice([0.0, 0.0, 1e3, 1e-3]))
    a = pkg.new_problem(lm_fused=FUSED, user_lambda_init=lam); a.upload_window(w)
    b = orc.new_problem(user_lambda_init=lam); b.upload_window(w)
    sa1, sb1 = a.optimize(4), b.optimize(4)
    ga, gb = a.gate_outliers(), b.gate_outliers()
    sa, sb = a.optimize(4), b.optimize(4)
    fused = int(a.debug_get("lm_fused")[0])
    ka, kb = a.get_keyframes(), b.get_keyframes()
    dP = np.abs(ka["P"] - kb["P"]).max(); dV = np.abs(ka["V"] - kb["V"]).max() if imu else 0.0
    dq = np.abs(ka["q"] - kb["q"]).max()
    dpt = np.abs(a.get_points() - b.get_points()).max() if Np else 0.0
    dln = np.abs(a.get_lines() - b.get_lines()).max() if Nl else 0.0
    # A small lambda_init is next to no damping: landmarks with a rank-deficient Hll (two observations from nearly the same ray, one left
    # after gating) make (Hll + lambda I) singular within rounding, and no two fp64 solvers agree on such a landmark — the record-based
    # passes deviate from the oracle on exactly the same cases and by as much (SOAK_LM_FUSED=0).  There: same decisions, chi2 and keyframes
    # to 1e-3; the count of failed factorisations (a pivot at the rounding level) is not compared.  (With lambda_init = 1e-6 even the
    # DECISIONS differ between all three — e.g. after gating the oracle and the record-based passes fail three factorisations where the fused
    # passes' Schur complement stays positive definite and its first step is accepted; every run ends at the same chi2.  Not drawn here.)
    loose = 0.0 < lam <= 1e-3
    same = (sa1.iterations, sa1.trials, sa.iterations, sa.trials) == (sb1.iterations, sb1.trials, sb.iterations, sb.trials) and ga == gb
    # at the fixed point a trial's gain is (chi2 - chi2') / scale with a numerator at the rounding level: whether the last trials are accepted
    # is noise.  Converged to the same state (1e-9) and the same chi2 (1e-10): the trial counts of the last call may differ.
    conv = ga == gb and (sa1.iterations, sa1.trials) == (sb1.iterations, sb1.trials) and max(dP, dV, dq) < 1e-9 and abs(sa.chi2_final - sb.chi2_final) <= 1e-10 * max(abs(sb.chi2_final), 1e-9)
    same = same or conv
    if not loose: same = same and (sa1.solver_failures + sa.solver_failures) == (sb1.solver_failures + sb.solver_failures)
    t = 1e3 if loose else 1.0
    ok = same and dP < 1e-6 * t and dV < 1e-5 * t and dq < 1e-6 * t and (loose or (dpt < 1e-5 and dln < 1e-5)) and abs(sa.chi2_final - sb.chi2_final) <= 1e-6 * t * max(abs(sb.chi2_final), 1e-9)
    verdict = "ok" if ok else "MISMATCH"
    nf_hip, nf_orc = sa1.solver_failures + sa.solver_failures, sb1.solver_failures + sb.solver_failures
    if not ok and nf_orc > nf_hip and max(dP, dV, dq) < 1e-1:
        # the fp64 ORACLE failed a factorisation the device did not (its Hpp - Hpl D Hpl^T cancels where the fused square-root form does not:
        # tests/test_fused_overshoot.py): the quad-precision build of the oracle arbitrates — trial counts and gating must be ITS
        try:
            q = orc.new_quad_problem(user_lambda_init=lam); q.upload_window(w)
            sq1 = q.optimize(4); gq = q.gate_outliers(); sq = q.optimize(4); kq = q.get_keyframes(); q.close()
            dq_quad = max(np.abs(ka[k] - kq[k]).max() for k in ("P", "q"))
            if (sa1.iterations, sa1.trials, sa.iterations, sa.trials) == (sq1.iterations, sq1.trials, sq.iterations, sq.trials) and ga == gq and sq1.solver_failures + sq.solver_failures == nf_hip and dq_quad < 1e-6:
                ok, verdict = True, "ok (the fp64 oracle failed %d factorisation(s); decisions = the quad-precision oracle's, |HIP - quad| %.1e)" % (nf_orc - nf_hip, dq_quad)
        except Exception as e:      # noqa: BLE001 — no quad build on this box: the mismatch stands
            verdict = "MISMATCH (quad arbiter unavailable: %s)" % e
    print("%3d K=%2d Np=%4d Nl=%4d tracks %d..%d imu %d lam %-6g %-22s fused %d groups %3d | trials %d+%d / %d+%d gated %s/%s chi2 %.6e / %.6e fails %d/%d dP %.1e dV %.1e dq %.1e dl %.1e %s" %
          (case, K, Np, Nl, lo, min(hi, K), imu, lam, "+".join(tag), fused, int(a.debug_get("lm_fused")[1]), sa1.trials, sa.trials, sb1.trials, sb.trials, ga, gb,
           sa.chi2_final, sb.chi2_final, nf_hip, nf_orc, dP, dV, dq, max(dpt, dln), verdict), flush=True)
    bad += not ok
    a.close(); b.close()
print("%d cases, %d mismatches" % (N, bad))
sys.exit(1 if bad else 0)
