"""BASELINE configs[3] and configs[4] at their real shape.

configs[4] = 200 KF / 200k points / 40k lines + IMU: the oracle needs minutes per iteration at that landmark count, so the
full size goes through size-independent properties (chi2 monotone over accepted steps, no solver failure, save / restore
replays bit for bit, the chain path and the dense path agree), and the SAME 200-keyframe window shape is compared with the
oracle at a landmark count it finishes in seconds.  configs[3] = the 50-keyframe window with a DEVICE-built marginalization
prior: chain path with prior-forced separators and several segments, against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pose_delta(a, b, pkg):
    dP = np.abs(a["P"] - b["P"]).max()
    dV = np.abs(a["V"] - b["V"]).max()
    dphi = 0.0
    for qa, qb in zip(a["q"], b["q"]):
        Ra, Rb = pkg.window.R_from_quat(qa), pkg.window.R_from_quat(qb)
        dphi = max(dphi, np.linalg.norm(pkg.window.log_so3(Rb.T @ Ra)))
    db = max(np.abs(a["dbg"] - b["dbg"]).max(), np.abs(a["dba"] - b["dba"]).max())
    return dP, dV, dphi, db


@pytest.fixture(scope="module")
def w5(pkg):
    return pkg.window.make_config(5)


def test_config5_full_size_properties(pkg, hip, w5):
    """BASELINE configs[4] at full size on one GPU through the reference protocol (5 + 10 iterations, gating)."""
    assert (w5["meta"]["K"], w5["meta"]["Np"], w5["meta"]["Nl"]) == (200, 200000, 40000)
    g = pkg.new_problem(); g.upload_window(w5)
    r = pkg.protocol.local_ba(g)
    assert r["stage1"].iterations == 5 and r["stage2"].iterations == 10
    assert r["stage1"].solver_failures == 0 and r["stage2"].solver_failures == 0
    assert g.debug_get("pose_dim")[0] == 199 * 15 and g.debug_get("dense_dim")[0] < 199 * 15      # chain path in effect
    assert g.debug_get("twin")[0] == 1 and g.debug_get("band")[0] == 0      # 44 tiles, band of 3: the two-ended multi-launch factorisation solves it
    tr = g.trace()
    chi = [t["chi2_current"] for t in tr] + [tr[-1]["chi2_trial"] if tr[-1]["accepted"] else tr[-1]["chi2_current"]]
    assert all(b <= a * (1 + 1e-12) for a, b in zip(chi[:-1], chi[1:])), "chi2 must not increase over accepted LM steps"
    assert r["stage2"].chi2_final < r["stage2"].chi2_initial < r["stage1"].chi2_initial
    # the estimate converges towards the generating trajectory (5 % outliers were gated)
    kf = g.get_keyframes()
    assert np.abs(kf["P"] - w5["truth"]["P"]).max() < 0.02
    frac = r["gated"][0] / w5["meta"]["Ep"]
    assert 0.03 < frac < 0.15, frac
    # replay from a saved state is bit-identical (deterministic reductions, no atomics on the landmark / Schur path)
    g.save_state()
    a = g.optimize(3); ka = g.get_keyframes(); pa = g.get_points()
    g.restore_state()
    b = g.optimize(3); kb = g.get_keyframes(); pb = g.get_points()
    assert a.chi2_final == b.chi2_final
    assert ka["P"].tobytes() == kb["P"].tobytes() and pa.tobytes() == pb.tobytes()
    g.close()


def test_config5_chain_path_matches_dense_path(pkg, hip, w5):
    """same full-size window: velocity / bias chain eliminated ahead of the dense factorisation (default) against the dense
    path on all 2985 pose dims (94 block steps, dataflow back-substitution)"""
    res = []
    for chain in (1, 0):
        g = pkg.new_problem(chain_elim=chain); g.upload_window(w5)
        st = g.optimize(3)
        assert st.solver_failures == 0 and st.iterations == 3
        res.append((st, g.get_keyframes(), g.get_points()[:2000].copy(), [t["accepted"] for t in g.trace()]))
        g.close()
    (sa, ka, pa, ta), (sb, kb, pb, tb) = res
    assert ta == tb
    assert sa.chi2_final == pytest.approx(sb.chi2_final, rel=1e-9)
    assert max(_pose_delta(ka, kb, pkg)) < 1e-7
    assert np.abs(pa - pb).max() < 1e-7


def test_config5_window_shape_against_the_oracle(pkg, orc, hip):
    """the 200-keyframe window (P = 2985 pose dims) with 6k points / 1.2k lines: two-stage-shaped run against the oracle"""
    w = pkg.window.make_window(200, 6000, 1200, imu=True, seed=0x5EED0005)
    g = pkg.new_problem(); g.upload_window(w)
    o = orc.new_problem(); o.upload_window(w)
    sg, so = g.optimize(2), o.optimize(2)
    assert (sg.iterations, sg.trials, sg.solver_failures) == (so.iterations, so.trials, 0)
    assert g.gate_outliers(pkg.window.CHI2_GATE) == o.gate_outliers(pkg.window.CHI2_GATE)
    sg, so = g.optimize(2), o.optimize(2)
    assert (sg.iterations, sg.trials, sg.solver_failures) == (so.iterations, so.trials, 0)
    assert sg.chi2_final == pytest.approx(so.chi2_final, rel=1e-9)
    assert max(_pose_delta(g.get_keyframes(), o.get_keyframes(), pkg)) < 1e-8
    assert np.abs(g.get_points() - o.get_points()).max() < 1e-7
    g.close(); o.close()


def test_config4_k50_device_prior_chain_path(pkg, orc, hip):
    """BASELINE configs[3]: the 50-keyframe IMU window carrying a marginalization prior BUILT ON THE DEVICE from the preceding
    BA of the same window.  The prior's kept vertices become forced separators of the chain elimination, which still runs
    with several segments (K = 50); compared with the oracle stage by stage and through the two-stage protocol."""
    w = pkg.window.make_config(3, scale=0.1)
    g = pkg.new_problem(); g.upload_window(w)
    pkg.protocol.local_ba(g)
    pr = g.marginalize(0, pkg.protocol.MARG_NUM)
    g.close()
    assert pr["n"] >= 15 and pkg.protocol.next_window_prior_ok(pr, w["kf"]["vid_pvr"], w["kf"]["vid_bias"])
    w2 = pkg.window.make_config(3, scale=0.1)
    w2["prior"] = pr
    g = pkg.new_problem(); g.upload_window(w2)
    o = orc.new_problem(); o.upload_window(w2)
    g.debug_build(5.0, False); o.debug_build(5.0, False)
    for name in ("err_prior", "bp", "bschur", "Hschur", "chi2", "maxdiag"):
        a, b = g.debug_get(name), o.debug_get(name)
        assert np.abs(a - b).max() <= 1e-9 * max(np.abs(b).max(), 1e-300), name
    rg, ro = pkg.protocol.local_ba(g), pkg.protocol.local_ba(o)
    pose_dim, dense_dim = g.debug_get("pose_dim")[0], g.debug_get("dense_dim")[0]
    assert dense_dim < pose_dim                                        # chain path in effect despite the forced separators
    forced = {int(v) // 2 for v in pr["vid"]}                          # keyframes the prior touches (vertex ids 2k, 2k + 1)
    assert dense_dim >= 49 * 6 + 9 * len(forced)                       # their velocity / bias dims stay in the dense system
    assert rg["gated"] == ro["gated"]
    assert (rg["stage2"].iterations, rg["stage2"].trials, rg["stage2"].solver_failures) == (ro["stage2"].iterations, ro["stage2"].trials, 0)
    assert rg["stage2"].chi2_final == pytest.approx(ro["stage2"].chi2_final, rel=1e-8)
    assert max(_pose_delta(g.get_keyframes(), o.get_keyframes(), pkg)) < 1e-7
    g.close(); o.close()


def test_config4_full_size_sliding_window(pkg, hip):
    """configs[3] at full size (50 KF / 20k / 4k): BA, device marginalization, next BA with the prior edge; properties only
    (the oracle's share of this configuration is test_config4_k50_device_prior_chain_path)"""
    w = pkg.window.make_config(4)
    g = pkg.new_problem(); g.upload_window(w)
    pkg.protocol.local_ba(g)
    pr = g.marginalize(0, pkg.protocol.MARG_NUM)
    g.close()
    A = pr["J0"].T @ pr["J0"]
    ev = np.linalg.eigvalsh(pr["Ar"])
    keep = ev > 1e-8
    assert np.abs(np.linalg.eigvalsh(A)[-keep.sum():] - ev[keep]).max() < 1e-7 * ev.max()      # J0^T J0 = thresholded A'
    w["prior"] = pr
    g = pkg.new_problem(); g.upload_window(w)
    r = pkg.protocol.local_ba(g)
    assert r["stage2"].solver_failures == 0 and r["stage2"].iterations == 10
    assert r["stage2"].chi2_final < r["stage1"].chi2_initial
    assert np.abs(g.get_keyframes()["P"] - w["truth"]["P"]).max() < 0.05
    g.close()


@pytest.mark.parametrize("K,Np,Nl", [(50, 2000, 400), (33, 900, 200), (100, 3000, 600)])
def test_banded_twisted_solver_matches_the_dense_path_and_the_oracle(pkg, orc, hip, K, Np, Nl):
    """plba_band.hip: when the compact dense system is block-banded (<= 3 sub-diagonal tiles: tracks over <= 8 consecutive
    keyframes) two workgroups factor it from both ends inside LDS.  Same optimisation as the dense multi-launch path
    (band_solve = 0) and as the oracle."""
    w = pkg.window.make_window(K, Np, Nl, imu=True, seed=0xBA4D + K)
    res = {}
    for band in (1, 0):
        g = pkg.new_problem(band_solve=2 * band); g.upload_window(w)      # 2: also below the 24-tile threshold of the default
        g.debug_build(3.0, True)
        x = g.debug_get("x").copy()
        assert g.debug_get("band")[0] == band and g.debug_get("solver_ok")[0] == 1
        st = g.optimize(4)
        res[band] = (x, st, g.get_keyframes(), [t["accepted"] for t in g.trace()])
        g.close()
    o = orc.new_problem(); o.upload_window(w)
    o.debug_build(3.0, True)
    xo = o.debug_get("x").copy()
    so = o.optimize(4)
    ko = o.get_keyframes()
    o.close()
    sc = np.abs(xo).max()
    assert np.abs(res[1][0] - xo).max() < 1e-7 * sc and np.abs(res[0][0] - xo).max() < 1e-7 * sc
    assert np.abs(res[1][0] - res[0][0]).max() < 1e-9 * sc
    for band in (1, 0):
        st = res[band][1]
        assert (st.iterations, st.trials, st.solver_failures) == (so.iterations, so.trials, 0)
        assert st.chi2_final == pytest.approx(so.chi2_final, rel=1e-8)
        assert max(_pose_delta(res[band][2], ko, pkg)) < 1e-8
    assert res[1][3] == res[0][3]


def test_banded_solver_reports_a_non_positive_pivot(pkg, hip):
    """a window whose damped system is not positive definite must come back as a failed solve from the band path too"""
    w = pkg.window.make_window(50, 1500, 300, imu=True, seed=0xBA4D)
    g = pkg.new_problem(band_solve=2); g.upload_window(w)
    g.debug_build(-1e13, True)               # lambda far below zero: indefinite on purpose
    assert g.debug_get("band")[0] == 1 and g.debug_get("solver_ok")[0] == 0
    g.close()


@pytest.mark.parametrize("K,Np,Nl", [(50, 2000, 400), (36, 900, 200), (80, 2500, 500), (200, 6000, 1200)])
def test_twin_factorisation_matches_the_dense_path_and_the_oracle(pkg, orc, hip, K, Np, Nl):
    """plba_dense.hip launch_twin_cholesky (the default for 8 <= tiles < 24): the banded compact system stored permuted
    [top chain | bottom chain reversed | middle], both ends eliminated side by side in each launch.  Same optimisation as the
    plain dense path (band_solve = 0) and as the oracle."""
    w = pkg.window.make_window(K, Np, Nl, imu=True, seed=0x7B1A + K)
    res = {}
    for twin in (1, 0):
        g = pkg.new_problem(band_solve=twin); g.upload_window(w)
        g.debug_build(3.0, True)
        x = g.debug_get("x").copy()
        assert g.debug_get("twin")[0] == twin and g.debug_get("band")[0] == 0 and g.debug_get("solver_ok")[0] == 1
        st = g.optimize(4)
        res[twin] = (x, st, g.get_keyframes(), [t["accepted"] for t in g.trace()])
        g.close()
    o = orc.new_problem(); o.upload_window(w)
    o.debug_build(3.0, True)
    xo = o.debug_get("x").copy()
    so = o.optimize(4)
    ko = o.get_keyframes()
    o.close()
    sc = np.abs(xo).max()
    assert np.abs(res[1][0] - xo).max() < 1e-7 * sc and np.abs(res[0][0] - xo).max() < 1e-7 * sc
    assert np.abs(res[1][0] - res[0][0]).max() < 1e-9 * sc
    for twin in (1, 0):
        st = res[twin][1]
        assert (st.iterations, st.trials, st.solver_failures) == (so.iterations, so.trials, 0)
        assert st.chi2_final == pytest.approx(so.chi2_final, rel=1e-8)
        assert max(_pose_delta(res[twin][2], ko, pkg)) < 1e-8
    assert res[1][3] == res[0][3]


def test_twin_factorisation_reports_a_non_positive_pivot(pkg, hip):
    w = pkg.window.make_window(50, 1500, 300, imu=True, seed=0xBA4D)
    g = pkg.new_problem(); g.upload_window(w)
    g.debug_build(-1e13, True)               # lambda far below zero: indefinite on purpose
    assert g.debug_get("twin")[0] == 1 and g.debug_get("solver_ok")[0] == 0
    g.close()


@pytest.mark.parametrize("K,seed,prior,fixed,lam0", [(30, 2, False, False, 0.0), (41, 5, False, True, 1e3), (64, 9, True, False, 0.0), (110, 14, False, False, 0.0),
                                                     (128, 15, True, False, 1e3), (170, 17, False, True, 0.0)])
def test_multi_chain_plans_against_the_oracle(pkg, orc, hip, K, seed, prior, fixed, lam0):
    """tools/soak_solver.py in small: window lengths that give two- and four-chain plans with separators of different widths
    (a marginalization prior widens the band), fixed keyframes, an overshooting lambda_init (rejected trials)"""
    w = pkg.window.make_window(K, 12 * K, 3 * K, imu=True, seed=0x50A0 + seed)
    if prior:
        p0 = pkg.new_problem(); p0.upload_window(w); pkg.protocol.local_ba(p0); pr = p0.marginalize(0, 50); p0.close()
        w = pkg.window.make_window(K, 12 * K, 3 * K, imu=True, seed=0x50A0 + seed); w["prior"] = pr
    if fixed:
        w["kf"]["fixed_pvr"] = np.zeros(K, np.uint8); w["kf"]["fixed_pvr"][:2] = 1
    g = pkg.new_problem(user_lambda_init=lam0); g.upload_window(w)
    o = orc.new_problem(user_lambda_init=lam0); o.upload_window(w)
    sg, so = g.optimize(5), o.optimize(5)
    assert g.debug_get("twin")[0] == 1
    assert (sg.iterations, sg.trials, sg.solver_failures) == (so.iterations, so.trials, so.solver_failures)
    assert sg.chi2_final == pytest.approx(so.chi2_final, rel=1e-8)
    assert max(_pose_delta(g.get_keyframes(), o.get_keyframes(), pkg)) < 1e-8
    g.close(); o.close()


def test_realistic_12_keyframe_window_against_the_oracle(pkg, orc, hip):
    """the reference's own window shape — 12 keyframes, tracks over most of the window (6 .. 12 keyframes), one
    landmark in five seen again after a gap — through the DEFAULT options: whatever solver / landmark path the structure detection
    picks (the band of the reduced system is as wide as the system here; no landmark group fits a window of 8 keyframes) must agree
    with the oracle through the whole protocol and through a marginalization slide."""
    w = pkg.window.make_window(12, 600, 120, imu=True, seed=0x5EED00C0, kf_dt=0.1, track=(6, 12), revisit=0.2)
    ks = np.diff(np.flatnonzero(np.diff(np.concatenate([[-1], w["po_pt"], [-2]])))); assert ks.max() > 8      # more observations per landmark than a group's window
    gaps = [np.diff(w["po_kf"][w["po_pt"] == l]).max() for l in range(0, 600, 7)]; assert max(gaps) > 1      # non-consecutive re-observations exist
    g = pkg.new_problem(lm_fused=2); g.upload_window(w)      # the fused passes take the window since round 4 (wide groups); default options: below
    o = orc.new_problem(); o.upload_window(w)
    rg, ro = pkg.protocol.local_ba(g), pkg.protocol.local_ba(o)
    assert g.debug_get("lm_fused")[0] == 1 and g.debug_get("lm_fused")[3] > 0      # ... with wide groups for the tracks over 9 .. 12 keyframes
    assert rg["gated"] == ro["gated"]
    assert [t["accepted"] for t in g.trace()] == [t["accepted"] for t in o.trace()]
    assert rg["stage2"].chi2_final == pytest.approx(ro["stage2"].chi2_final, rel=1e-9)
    assert max(_pose_delta(g.get_keyframes(), o.get_keyframes(), pkg)) < 1e-9
    pg, po = g.marginalize(0, pkg.protocol.MARG_NUM), o.marginalize(0, pkg.protocol.MARG_NUM)
    assert pg["n"] == po["n"] and pg["n"] > 100      # every keyframe of the window among the kept parameters
    assert np.abs(pg["Ar"] - po["Ar"]).max() < 1e-7 * np.abs(po["Ar"]).max()
    print("realistic window: dense dim %d of %d, twin %d band %d" % (g.debug_get("dense_dim")[0], g.debug_get("pose_dim")[0], g.debug_get("twin")[0], g.debug_get("band")[0]))
    g.close()
    d = pkg.new_problem(); d.upload_window(w)      # DEFAULT options: whichever landmark path the size threshold picks, the same results
    rd = pkg.protocol.local_ba(d)
    assert rd["gated"] == ro["gated"] and rd["stage2"].chi2_final == pytest.approx(ro["stage2"].chi2_final, rel=1e-9)
    assert max(_pose_delta(d.get_keyframes(), o.get_keyframes(), pkg)) < 1e-9
    d.close(); o.close()


def test_long_track_50_keyframe_window_against_the_oracle(pkg, orc, hip):
    """a 50-keyframe window whose tracks span 2 .. 15 keyframes, one landmark in five seen again up to 30
    keyframes later — no usable band in the reduced camera system (plain dense factorisation), standard and wide landmark groups side by side — through
    the whole protocol with default options and with the fused passes forced, against the oracle.  (bench.py times the full-size window
    of this shape as config.long_track_50kf_window.)"""
    w = pkg.window.make_window(50, 4000, 800, imu=True, seed=0x5EED00D0, track=(2, 15), revisit=0.2, revisit_gap=(1, 30))
    ks = np.diff(np.flatnonzero(np.diff(np.concatenate([[-1], w["po_pt"], [-2]])))); assert 8 < ks.max() <= 16
    span = max(int(np.ptp(w["po_kf"][w["po_pt"] == l])) for l in range(0, 4000, 25)); assert span >= 20      # keyframes 20+ apart share landmarks: no band
    o = orc.new_problem(); o.upload_window(w)
    ro = pkg.protocol.local_ba(o)
    for opts in (dict(), dict(lm_fused=2)):
        g = pkg.new_problem(**opts); g.upload_window(w)
        rg = pkg.protocol.local_ba(g)
        if opts: assert g.debug_get("lm_fused")[0] == 1 and g.debug_get("lm_fused")[3] > 0
        print("long-track window %s: dense dim %d, twin %d band %d, pose-pair blocks %d" % (opts, g.debug_get("dense_dim")[0], g.debug_get("twin")[0], g.debug_get("band")[0], g.debug_get("lm_fused")[2]))
        assert rg["gated"] == ro["gated"]
        assert [t["accepted"] for t in g.trace()] == [t["accepted"] for t in o.trace()]
        assert rg["stage2"].chi2_final == pytest.approx(ro["stage2"].chi2_final, rel=1e-9)
        assert max(_pose_delta(g.get_keyframes(), o.get_keyframes(), pkg)) < 1e-9
        g.close()
    o.close()
