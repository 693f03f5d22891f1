"""GPU parity tests: every stage of the HIP hot path against the CPU oracle on identical seeded
windows, through the C ABI (include/plba.h).  Tolerances are fp64 rounding scaled by magnitude."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-5      # BASELINE.json north_star: final pose deltas within 1e-5 of the reference path


def _pair(pkg, orc, w, **opts):
    g = pkg.new_problem(**opts); g.upload_window(w)
    o = orc.new_problem(**opts); o.upload_window(w)
    return g, o


def _close(a, b, rtol=1e-9, name=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    sc = max(np.abs(b).max(), 1e-300) if b.size else 1.0
    err = np.abs(a - b).max() / sc if b.size else 0.0
    assert err < rtol, "%s: max rel err %.3e (scale %.3e)" % (name, err, sc)


def _pose_delta(a, b, pkg):
    dP = np.abs(a["P"] - b["P"]).max()
    dV = np.abs(a["V"] - b["V"]).max()
    dphi = 0.0
    for qa, qb in zip(a["q"], b["q"]):
        Ra, Rb = pkg.window.R_from_quat(qa), pkg.window.R_from_quat(qb)
        dphi = max(dphi, np.linalg.norm(pkg.window.log_so3(Rb.T @ Ra)))
    db = max(np.abs(a["dbg"] - b["dbg"]).max(), np.abs(a["dba"] - b["dba"]).max())
    return dP, dV, dphi, db


def test_backend_is_the_hip_library(hip):
    assert hip.backend_name() == "hip-gfx950"


@pytest.mark.parametrize("n", [1, 5, 63, 64, 65, 200, 750])
@pytest.mark.parametrize("mfma,fb,flow,wide", [(1, 32, 0, 1), (1, 32, 0, 0), (1, 32, 1, 0), (0, 32, 0, 0), (1, 64, 0, 0), (0, 64, 0, 0)])
def test_dense_solver_vs_numpy(pkg, hip, n, mfma, fb, flow, wide):
    """wide = 0 (default): one launch per 32 columns; wide = 1: 64 columns per launch (two pipelined 32-column sweeps in the
    look-ahead workgroup); flow = 1: the single-launch dataflow factorisation; mfma = 0 / fb = 64: the VALU check paths"""
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n)); A = A @ A.T + n * np.eye(n)
    b = rng.normal(size=n)
    p = pkg.new_problem(use_mfma=mfma, factor_block=fb, factor_flow=flow, wide_steps=wide)
    x, ok = p.debug_dense_solve(A, b)
    assert ok
    _close(x, np.linalg.solve(A, b), 1e-9, "x")
    # a non positive definite matrix must be reported, not silently solved
    A[n // 2, n // 2] = -1.0
    _, ok = p.debug_dense_solve(A, b)
    assert not ok
    p.close()


@pytest.mark.parametrize("imu", [True, False])
def test_edge_errors_and_depth(pkg, orc, hip, imu):
    w = pkg.window.make_window(8, 300, 70, imu=imu, seed=101)
    g, o = _pair(pkg, orc, w)
    g.recompute_errors(); o.recompute_errors()
    for kind in (pkg.abi.EDGE_POINT, pkg.abi.EDGE_LINE) + ((pkg.abi.EDGE_IMU_PVR, pkg.abi.EDGE_IMU_BIAS) if imu else ()):
        cg, dg = g.edge_chi2(kind); co, do = o.edge_chi2(kind)
        _close(cg, co, 1e-9, "chi2 kind %d" % kind)
        assert np.array_equal(dg, do)
    g.close(); o.close()


@pytest.mark.parametrize("imu", [True, False])
def test_build_system_and_schur(pkg, orc, hip, imu):
    w = pkg.window.make_window(7, 200, 50, imu=imu, seed=102)
    g, o = _pair(pkg, orc, w)
    lam = 12.5
    g.debug_build(lam, False); o.debug_build(lam, False)
    assert g.debug_get("pose_dim")[0] == o.debug_get("pose_dim")[0]
    for name in ("err_pt", "err_ln", "hll_pt", "bl_pt", "hll_ln", "bl_ln", "chi2", "maxdiag", "bp", "bschur", "Hschur") + \
            (("err_pvr", "err_bias") if imu else ()):
        _close(g.debug_get(name), o.debug_get(name), 1e-9, name)
    g.close(); o.close()


@pytest.mark.parametrize("mfma,fb", [(1, 32), (0, 32), (1, 64)])
def test_one_damped_solve(pkg, orc, hip, mfma, fb):
    w = pkg.window.make_window(7, 200, 50, imu=True, seed=103)
    g, o = _pair(pkg, orc, w, use_mfma=mfma, factor_block=fb)
    g.debug_build(3.0, True); o.debug_build(3.0, True)
    assert g.debug_get("solver_ok")[0] == 1
    _close(g.debug_get("x"), o.debug_get("x"), 1e-7, "x")
    g.close(); o.close()


def test_lm_trace_and_final_state(pkg, orc, hip):
    w = pkg.window.make_window(10, 400, 80, imu=True, seed=104)
    g, o = _pair(pkg, orc, w)
    sg, so = g.optimize(5), o.optimize(5)
    assert (sg.iterations, sg.trials, sg.stop_reason) == (so.iterations, so.trials, so.stop_reason)
    tg, to = g.trace(), o.trace()
    assert len(tg) == len(to)
    for a, b in zip(tg, to):
        assert (a["iteration"], a["trial"], a["accepted"], a["solver_ok"]) == (b["iteration"], b["trial"], b["accepted"], b["solver_ok"])
        for k in ("lam", "chi2_current", "chi2_trial", "scale"):
            assert a[k] == pytest.approx(b[k], rel=1e-7), k
    kg, ko = g.get_keyframes(), o.get_keyframes()
    assert max(_pose_delta(kg, ko, pkg)) < 1e-8
    _close(g.get_points(), o.get_points(), 1e-8, "points")
    _close(g.get_lines(), o.get_lines(), 1e-8, "lines")
    assert sg.chi2_final == pytest.approx(so.chi2_final, rel=1e-8)
    g.close(); o.close()


def _two_stage(pkg, orc, w, tol=POSE_TOL):
    g, o = _pair(pkg, orc, w)
    rg, ro = pkg.protocol.local_ba(g), pkg.protocol.local_ba(o)
    assert rg["gated"] == ro["gated"]
    assert np.array_equal(g.get_levels(pkg.abi.EDGE_POINT), o.get_levels(pkg.abi.EDGE_POINT))
    assert np.array_equal(g.get_levels(pkg.abi.EDGE_LINE), o.get_levels(pkg.abi.EDGE_LINE))
    assert rg["stage2"].iterations == ro["stage2"].iterations and rg["stage2"].trials == ro["stage2"].trials
    d = _pose_delta(g.get_keyframes(), o.get_keyframes(), pkg)
    assert max(d) < tol, d
    assert np.abs(g.get_points() - o.get_points()).max() < 10 * tol
    assert np.abs(g.get_lines() - o.get_lines()).max() < 10 * tol
    assert rg["stage2"].chi2_final == pytest.approx(ro["stage2"].chi2_final, rel=1e-6)
    g.close(); o.close()
    return d


def test_two_stage_protocol_small(pkg, orc, hip):
    _two_stage(pkg, orc, pkg.window.make_window(12, 500, 100, imu=True, seed=105))


def test_two_stage_protocol_with_the_single_launch_factorisation(pkg, orc, hip):
    """factor_flow = 1 (k_chol_flow, the experimental dataflow factorisation) gives the same optimisation"""
    w = pkg.window.make_window(12, 500, 100, imu=True, seed=105)
    g, o = _pair(pkg, orc, w, factor_flow=1)
    rg, ro = pkg.protocol.local_ba(g), pkg.protocol.local_ba(o)
    assert rg["gated"] == ro["gated"] and rg["stage2"].solver_failures == 0
    assert max(_pose_delta(g.get_keyframes(), o.get_keyframes(), pkg)) < POSE_TOL
    g.close(); o.close()


def test_config1_full_size_no_imu(pkg, orc, hip):
    """BASELINE configs[0]: 10 KF / 2k points / 500 lines, no IMU, no marg."""
    _two_stage(pkg, orc, pkg.window.make_config(1))


def test_config2_no_imu_reduced(pkg, orc, hip):
    _two_stage(pkg, orc, pkg.window.make_config(2, scale=0.2))


def test_config3_imu_reduced(pkg, orc, hip):
    _two_stage(pkg, orc, pkg.window.make_config(3, scale=0.1))


@pytest.mark.parametrize("K,Np,Nl,imu", [(12, 500, 100, True), (100, 1500, 300, True), (9, 200, 40, False), (2, 30, 6, True)])
@pytest.mark.parametrize("chain", [0, 1])
def test_chain_variable_elimination_option(pkg, orc, hip, K, Np, Nl, imu, chain):
    """chain_elim = 1 (default, plba_chain.hip): velocity / bias variables eliminated segment-wise ahead of the dense
    factorisation; chain_elim = 0: the dense path on the full system.  Both give the oracle's optimisation."""
    w = pkg.window.make_window(K, Np, Nl, imu=imu, seed=0xC4A1 + K)
    g, o = _pair(pkg, orc, w, chain_elim=chain)
    sg, so = g.optimize(4), o.optimize(4)
    assert (g.debug_get("dense_dim")[0] < g.debug_get("pose_dim")[0]) == bool(chain)          # the option is really in effect
    assert (sg.iterations, sg.trials, sg.solver_failures) == (so.iterations, so.trials, 0)
    assert sg.chi2_final == pytest.approx(so.chi2_final, rel=1e-7)
    assert max(_pose_delta(g.get_keyframes(), o.get_keyframes(), pkg)) < (1e-7 if imu else 1e-4)
    g.close(); o.close()


def _variant(pkg, w, kind):
    """structural edge cases for the chain elimination / index maps, applied to a generated window"""
    K = len(w["kf"]["P"])
    kf = w["kf"]
    if kind == "fixed_middle":           # a fixed keyframe inside the window: its chain block is missing, two chains
        kf["fixed_pvr"] = kf["fixed_pvr"].copy(); kf["fixed_bias"] = kf["fixed_bias"].copy()
        kf["fixed_pvr"][K // 2] = 1; kf["fixed_bias"][K // 2] = 1
    elif kind == "fixed_bias_only":      # pose free, bias fixed: chain block with 3 live dims
        kf["fixed_bias"] = kf["fixed_bias"].copy(); kf["fixed_bias"][[2, K - 2]] = 1
    elif kind == "fixed_pose_only":      # bias free, pose/velocity fixed: chain block with 6 live dims, no pose block
        kf["fixed_pvr"] = kf["fixed_pvr"].copy(); kf["fixed_pvr"][3] = 1
    elif kind == "dropped_imu_edge":     # no IMU edge between two neighbours: the chain falls apart there
        im = dict(w["imu"]); keep = np.ones(K - 1, bool); keep[K // 3] = False
        for k in ("kf_i", "kf_j", "preint", "info_pvr", "info_bias"): im[k] = im[k][keep]
        w["imu"] = im
    elif kind == "nothing_fixed":        # gauge left to the damping
        kf["fixed_pvr"] = np.zeros(K, np.uint8); kf["fixed_bias"] = np.zeros(K, np.uint8)
    return w


@pytest.mark.parametrize("K", [9, 10, 17, 18, 19, 28])
@pytest.mark.parametrize("kind", ["plain", "fixed_middle", "fixed_bias_only", "fixed_pose_only", "dropped_imu_edge", "nothing_fixed"])
def test_window_structure_variants(pkg, orc, hip, K, kind):
    """segment-boundary window sizes x structural variants: whatever the index maps (positions, separators, segments, column
    windows, keyframe owners) come out as, the device takes the oracle's steps"""
    if kind != "plain" and K in (10, 18):
        pytest.skip("variant covered at the neighbouring sizes")
    w = _variant(pkg, pkg.window.make_window(K, 40 * K, 8 * K, imu=True, seed=0x57A0 + K), kind)
    g, o = _pair(pkg, orc, w)
    sg, so = g.optimize(4), o.optimize(4)
    assert (sg.iterations, sg.trials, sg.solver_failures) == (so.iterations, so.trials, so.solver_failures)
    assert sg.chi2_final == pytest.approx(so.chi2_final, rel=1e-7)
    assert max(_pose_delta(g.get_keyframes(), o.get_keyframes(), pkg)) < 1e-7
    g.close(); o.close()


def test_large_window_many_block_steps(pkg, orc, hip):
    """100 keyframes: P = 1485 pose dimensions, 47 block steps of the dense factorisation, 24 dataflow hops of the
    back-substitution — the shape class of BASELINE configs[4], on a landmark count the oracle finishes in seconds"""
    w = pkg.window.make_window(100, 3000, 600, imu=True, seed=0x5EED0005)
    g, o = _pair(pkg, orc, w)
    sg, so = g.optimize(3), o.optimize(3)
    assert (sg.iterations, sg.trials, sg.solver_failures) == (so.iterations, so.trials, 0)
    assert sg.chi2_final == pytest.approx(so.chi2_final, rel=1e-9)
    assert max(_pose_delta(g.get_keyframes(), o.get_keyframes(), pkg)) < 1e-9
    g.close(); o.close()


@pytest.mark.parametrize("K,Np,Nl,imu", [
    (2, 30, 6, True),        # the smallest IMU window: one free keyframe, one block step
    (6, 80, 0, True),        # points only
    (6, 0, 40, True),        # lines only
    (5, 0, 0, True),         # no observations at all: IMU chain + LM damping only
    (3, 40, 10, False),      # smallest no-IMU window (velocity held by damping alone)
    (9, 9, 3, True),         # almost every landmark seen by very few keyframes (rank-deficient landmark blocks)
])
def test_degenerate_windows(pkg, orc, hip, K, Np, Nl, imu):
    """ragged / empty inputs: whatever the oracle does with them, the device does the same"""
    w = pkg.window.make_window(K, Np, Nl, imu=imu, seed=0xDE6E + K + Np)
    g, o = _pair(pkg, orc, w)
    sg, so = g.optimize(4), o.optimize(4)
    assert (sg.iterations, sg.trials, sg.stop_reason, sg.solver_failures) == (so.iterations, so.trials, so.stop_reason, so.solver_failures)
    if np.isfinite(so.chi2_final):
        assert sg.chi2_final == pytest.approx(so.chi2_final, rel=1e-6, abs=1e-9)
    d = _pose_delta(g.get_keyframes(), o.get_keyframes(), pkg)
    assert max(d) < (1e-6 if imu else 1e-4), d          # no-IMU windows are gauge-deficient: conditioned by the damping only
    if Np:
        assert np.abs(g.get_points() - o.get_points()).max() < 1e-5
    g.close(); o.close()


def test_levels_and_inactive_landmarks(pkg, orc, hip):
    w = pkg.window.make_window(6, 120, 30, imu=True, seed=106)
    g, o = _pair(pkg, orc, w)
    lv = np.zeros(len(w["po_pt"]), np.uint8); lv[(w["po_pt"] == 3) | (w["po_pt"] % 7 == 0)] = 1
    ll = np.zeros(len(w["lo_ln"]), np.uint8); ll[w["lo_ln"] == 2] = 1
    for p in (g, o):
        p.set_levels(pkg.abi.EDGE_POINT, lv); p.set_levels(pkg.abi.EDGE_LINE, ll)
    sg, so = g.optimize(3), o.optimize(3)
    assert sg.chi2_final == pytest.approx(so.chi2_final, rel=1e-8)
    _close(g.get_points(), o.get_points(), 1e-8, "points")
    assert np.array_equal(g.get_points()[3], w["points"][3])
    g.close(); o.close()


def test_rejected_trials_follow_the_oracle(pkg, orc, hip):
    w = pkg.window.make_window(5, 60, 10, imu=False, seed=13)
    w["points"] = w["points"] + np.random.default_rng(0).normal(size=w["points"].shape) * 1.5
    g, o = _pair(pkg, orc, w, user_lambda_init=1e-3)
    sg, so = g.optimize(4), o.optimize(4)
    tg, to = g.trace(), o.trace()
    # the first trial sees identical inputs: same decision, same numbers
    assert tg[0]["accepted"] == to[0]["accepted"]
    assert tg[0]["chi2_trial"] == pytest.approx(to[0]["chi2_trial"], rel=1e-3)   # gauge-deficient (no IMU, tiny lambda): cond ~1e10
    assert any(not r["accepted"] for r in tg) == any(not r["accepted"] for r in to)
    # rejected steps restore the state and raise lambda by nu = 2, 4, 8 ...
    for a, b in zip(tg[:-1], tg[1:]):
        if not a["accepted"] and b["iteration"] == a["iteration"]:
            assert b["lam"] > a["lam"] and b["chi2_current"] == a["chi2_current"]
    if [r["accepted"] for r in tg] == [r["accepted"] for r in to]:
        assert sg.chi2_final == pytest.approx(so.chi2_final, rel=1e-3)
    assert sg.chi2_final <= sg.chi2_initial
    g.close(); o.close()


def test_observation_culling_flags_match_the_oracle(pkg, orc, hip):
    """the culling decision the call site takes after the final optimize (mapHandler.cpp:5541-5620):
    level-1 edges re-evaluated on the final estimates, chi2 > 5.991 || !isDepthPositive => bad.  Bit-exact flags."""
    w = pkg.window.make_window(10, 400, 80, imu=True, seed=0xC011)
    rng = np.random.default_rng(5)
    w["po_uv"] = w["po_uv"].copy()
    bad = rng.choice(len(w["po_uv"]), 40, replace=False)
    w["po_uv"][bad] += rng.normal(size=(40, 2)) * 6.0            # gross outliers: gated after stage 1, then culled
    g, o = _pair(pkg, orc, w)
    pkg.protocol.local_ba(g); pkg.protocol.local_ba(o)
    assert np.array_equal(g.get_levels(pkg.abi.EDGE_POINT), o.get_levels(pkg.abi.EDGE_POINT))
    assert g.get_levels(pkg.abi.EDGE_POINT).sum() > 0
    cg, co = g.cull_observations(), o.cull_observations()
    assert np.array_equal(cg["bad_points"], co["bad_points"]) and np.array_equal(cg["bad_lines"], co["bad_lines"])
    assert (cg["n_points"], cg["n_lines"]) == (co["n_points"], co["n_lines"]) == (cg["bad_points"].sum(), cg["bad_lines"].sum())
    assert cg["n_points"] >= 30                                  # the planted outliers are among them
    # level-1 edges had their cached chi2 refreshed on the final estimates (computeError), like the reference's
    chg, _ = g.edge_chi2(pkg.abi.EDGE_POINT); cho, _ = o.edge_chi2(pkg.abi.EDGE_POINT)
    _close(chg, cho, 1e-9, "chi2 after culling")
    g.close(); o.close()


def test_rejected_trials_with_imu_edges(pkg, orc, hip):
    """retries on the default path (chain elimination, queued-ahead linearisation gated on the device-side decision, double
    buffered IMU accumulators): a rejected step must leave records and accumulators of the current state untouched.
    lambda = 1e-6 on landmarks perturbed by a metre: the damped systems are conditioned ~1e10 and fp64 solvers agree to a few digits
    only (first trial: oracle 588843, record-based passes 588734, fused passes 588718.5, EXACT 588718.7), so the values are held to the
    QUAD-precision run of the same window (tests/golden/fused_overshoot_quad.json, case rejected_small) with the fp64 oracle's own
    distance from it as the yardstick; what must agree exactly is the control flow.  (Round 3 compared against the fp64 oracle and
    the two device paths with each other at 1e-7: under the forced-fused suite that failed — the fused passes are the MORE accurate
    of the three, their square-root form of the Schur complement does not cancel.)"""
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fused_overshoot_quad.json")) as f:
        fx = json.load(f)["rejected_small"]
    w = pkg.window.make_window(10, 150, 30, imu=True, seed=77)
    w["points"] = w["points"] + np.random.default_rng(1).normal(size=w["points"].shape) * 1.0
    assert (int(w["meta"]["Ep"]), int(w["meta"]["El"])) == (fx["meta"]["Ep"], fx["meta"]["El"])
    q, o = fx["trace"], fx["trace_fp64_oracle"]
    dec = lambda tr: [(r["iteration"], r["trial"], r["accepted"]) for r in tr]
    assert any(not r["accepted"] for r in q), "the scenario is meant to produce rejected trials"

    def far(tr):      # largest relative distance of lambda / chi2 from the exact run
        return max(abs(a[k] - b[k]) / abs(b[k]) for a, b in zip(tr, q) for k in ("lam", "chi2_current", "chi2_trial"))
    yard = far(o)
    for opts in (dict(), dict(chain_elim=0)):
        g = pkg.new_problem(user_lambda_init=1e-6, **opts); g.upload_window(w)
        sg = g.optimize(6)
        tg = g.trace()
        assert dec(tg) == dec(q), opts
        assert tg[0]["chi2_current"] == pytest.approx(q[0]["chi2_current"], rel=1e-10)      # same start
        assert far(tg) <= 4 * yard, (opts, far(tg), yard)
        assert abs(tg[0]["chi2_trial"] - q[0]["chi2_trial"]) <= abs(o[0]["chi2_trial"] - q[0]["chi2_trial"])      # the first damped solve: no worse than the CPU fp64 solver
        # rejected steps restore the state (chi2_current unchanged) and raise lambda
        for a, b in zip(tg[:-1], tg[1:]):
            if not a["accepted"] and b["iteration"] == a["iteration"]:
                assert b["lam"] > a["lam"] and b["chi2_current"] == a["chi2_current"]
        assert sg.chi2_final <= sg.chi2_initial
        kf = g.get_keyframes()
        assert max(float(np.abs(kf[k] - np.asarray(fx["kf"][k])).max()) for k in fx["kf"]) <= 4 * max(fx["fp64_oracle_abs_diff"].values())
        g.close()


def _overshoot_window(pkg, seed, **kw):
    import os, sys
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    if here not in sys.path: sys.path.insert(0, here)
    import overshoot_cases
    return overshoot_cases.overshoot_window(pkg, seed, **kw)


def _overshoot_quad(name):
    """tests/golden/overshoot_quad.json: the window run through the quad-precision build of the oracle (make_overshoot_quad.py)"""
    import json, os
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "overshoot_quad.json")))[name]
    return {k: np.asarray(v) for k, v in d["kf"].items()}, d


def _trace_key(tr):
    return [(r["iteration"], r["trial"], r["accepted"], r["solver_ok"], r["lam"], r["chi2_current"], r["chi2_trial"], r["scale"]) for r in tr]


@pytest.mark.parametrize("chain", [1, 0])
def test_rejected_trials_against_the_oracle_on_an_overshooting_imu_window(pkg, orc, hip, chain):
    """VERDICT r01 weak #2.  lambda_init = 1e4 on an IMU window (2e10 on the diagonal: cond ~ 1e6..1e8), genuine overshoots
    rejected twice in a row at iteration 7 (chi2 7.97e5 -> 1.07e6, -> 8.81e5, then accepted).  Such a trajectory AMPLIFIES
    rounding: the oracle run twice with the points scaled by (1 + 1e-13) differs from itself by 8.7e-6 in chi2 at the
    rejected trials (tools/debug_rej.py), so the values that can be pinned against a different fp64 solver are: identical
    decisions, accepted steps to 1e-6, rejected overshoots to 1e-4, final states to 5e-5.  The bit-level check
    of the rejected-trial machinery itself is test_default_path_equals_synchronous_path_through_rejections."""
    w = _overshoot_window(pkg, 81)
    g, o = _pair(pkg, orc, w, user_lambda_init=1e4, chain_elim=chain)
    sg, so = g.optimize(8), o.optimize(8)
    tg, to = g.trace(), o.trace()
    rej = [not r["accepted"] for r in to]
    assert sum(rej) >= 2 and any(a and b for a, b in zip(rej[:-1], rej[1:])), "the scenario is meant to reject trials in a row"
    assert [(r["iteration"], r["trial"], r["accepted"], r["solver_ok"]) for r in tg] == [(r["iteration"], r["trial"], r["accepted"], r["solver_ok"]) for r in to]
    for a, b in zip(tg, to):
        tol = 1e-6 if b["accepted"] else 1e-4
        assert a["lam"] == pytest.approx(b["lam"], rel=1e-5) and a["chi2_current"] == pytest.approx(b["chi2_current"], rel=1e-6)
        assert a["chi2_trial"] == pytest.approx(b["chi2_trial"], rel=tol), (a, b)
    assert sg.chi2_final == pytest.approx(so.chi2_final, rel=1e-6)
    # Arbitration: how far apart may two correct fp64 solvers end on THIS window?  The oracle against itself with
    # its inputs moved by four ulps (points and pixel measurements scaled by 1 + 2^-50) — a perturbation below anything either
    # implementation controls.  Eight overshooting iterations from 20 m/s of velocity error amplify it to ~1e-5 on the final states;
    # the HIP path must stay within 4x of that self-sensitivity (and of the 1e-9 floor), which is the data behind the tolerance
    # that exceeds north_star's 1e-5 here and only here (the BASELINE windows agree to 1e-13: tests/test_golden_full.py).
    w2 = dict(w); w2["points"] = w["points"] * (1.0 + 2.0 ** -50); w2["po_uv"] = w["po_uv"] * (1.0 + 2.0 ** -50)
    o2 = orc.new_problem(user_lambda_init=1e4, chain_elim=chain); o2.upload_window(w2); o2.optimize(8)
    d_self = max(_pose_delta(o2.get_keyframes(), o.get_keyframes(), pkg))
    d_hip = max(_pose_delta(g.get_keyframes(), o.get_keyframes(), pkg))
    print("overshoot window: |HIP - oracle| = %.2e, |oracle(inputs + 4 ulp) - oracle| = %.2e" % (d_hip, d_self))
    assert d_hip < max(4.0 * d_self, 1e-9) and d_hip < 5e-5, (d_hip, d_self)
    # ... and against the QUAD-precision run of the same algorithm (tests/golden/overshoot_quad.json, generated by the
    # __float128 build of the oracle): the HIP path may be no further from it than 4 x the fp64 oracle is — both are fp64 solvers whose
    # rounding this trajectory amplifies by ~1e11 — and takes the quad run's decisions
    kq, fq = _overshoot_quad("seed81")
    d_hip_q = max(_pose_delta(g.get_keyframes(), kq, pkg))
    d_orc_q = max(_pose_delta(o.get_keyframes(), kq, pkg))
    print("                  |HIP - quad| = %.2e, |oracle64 - quad| = %.2e" % (d_hip_q, d_orc_q))
    assert d_hip_q <= 4.0 * d_orc_q + 1e-9, (d_hip_q, d_orc_q)
    assert [(r["iteration"], r["trial"], r["accepted"]) for r in tg] == [(r["iteration"], r["trial"], r["accepted"]) for r in fq["trace"]]
    g.close(); o.close(); o2.close()


@pytest.mark.parametrize("chain", [1, 0])
@pytest.mark.parametrize("seed", [77, 81])
def test_default_path_equals_synchronous_path_through_rejections(pkg, hip, seed, chain):
    """the default path (next linearisation queued ahead and gated on the device-side decision, IMU accumulators swapped on
    accept, LM decision taken by the last workgroup of the trial-error launch, mailbox polling) against the fully synchronous
    path (profile = 2: no speculation, a k_decide launch and a stream synchronisation per trial) on windows that reject steps at
    iterations 1-3 (seed 77) and 7 (seed 81): same kernels on the same data, so traces and states agree BIT FOR BIT — a record,
    accumulator or estimate left behind by a rejected trial would show here whatever the conditioning."""
    w = _overshoot_window(pkg, seed)
    res = []
    for prof in (0, 2):
        g = pkg.new_problem(user_lambda_init=1e4, chain_elim=chain, profile=prof); g.upload_window(w)
        st = g.optimize(8)
        res.append((_trace_key(g.trace()), g.get_keyframes(), g.get_points(), g.get_lines(), st.trials))
        g.close()
    (ta, ka, pa, la, na), (tb, kb, pb, lb, nb) = res
    assert any(not r[2] for r in ta), "the scenario is meant to reject trials"
    assert ta == tb and na == nb
    for k in ("P", "V", "q", "dbg", "dba"):
        assert ka[k].tobytes() == kb[k].tobytes(), k
    assert pa.tobytes() == pb.tobytes() and la.tobytes() == lb.tobytes()


def test_rejections_at_the_first_iteration_leave_no_trace(pkg, hip):
    """run A starts with lambda_init = 1e-3, overshoots and is rejected several times at iteration 0 until lambda has grown to
    lambda_k; run B starts at lambda_k directly.  g2o's pop() restores the estimates and the next trial re-damps the SAME
    linearisation, so from the accepted trial on the two runs must be the same computation: bit-equal traces and states."""
    w = _overshoot_window(pkg, 78, rot=0.4, vel=3.0, pts=1.0)
    a = pkg.new_problem(user_lambda_init=1e-3); a.upload_window(w)
    a.optimize(3)
    ta = a.trace()
    first_acc = next(i for i, r in enumerate(ta) if r["accepted"])
    assert first_acc >= 2 and ta[first_acc]["iteration"] == 0, "the scenario is meant to reject >= 2 trials at iteration 0"
    b = pkg.new_problem(user_lambda_init=ta[first_acc]["lam"]); b.upload_window(w)
    b.optimize(3)
    tb = b.trace()

    def key(tr):       # the trial counter of iteration 0 differs by construction
        return [(r["iteration"], r["accepted"], r["lam"], r["chi2_current"], r["chi2_trial"], r["scale"]) for r in tr]
    assert key(ta[first_acc:]) == key(tb)
    ka, kb = a.get_keyframes(), b.get_keyframes()
    for k in ("P", "V", "q", "dbg", "dba"):
        assert ka[k].tobytes() == kb[k].tobytes(), k
    assert a.get_points().tobytes() == b.get_points().tobytes()
    a.close(); b.close()


@pytest.mark.parametrize("chain", [1, 0])
def test_prior_edge_parity(pkg, orc, hip, chain):
    """BA with a marginalization prior: the oracle's prior on both sides (SURVEY B-Q3 decision).  With chain_elim = 1 the
    keyframes the prior touches keep their velocity / bias dims in the dense system (forced separators)."""
    w0 = pkg.window.make_window(12, 260, 50, imu=True, seed=22)
    o = orc.new_problem(); o.upload_window(w0)
    pkg.protocol.local_ba(o)
    pr = o.marginalize(0, 50)
    o.close()
    w0["prior"] = pr        # kept vertices all exist in w0 as well; KF0 itself is not among them
    w0["kf"]["fixed_pvr"] = np.zeros(12, np.uint8); w0["kf"]["fixed_pvr"][0] = 1
    g, o = _pair(pkg, orc, w0, chain_elim=chain)
    g.debug_build(5.0, False); o.debug_build(5.0, False)
    for name in ("err_prior", "bp", "bschur", "Hschur", "chi2", "maxdiag"):
        _close(g.debug_get(name), o.debug_get(name), 1e-9, name)
    sg, so = g.optimize(4), o.optimize(4)
    assert (g.debug_get("dense_dim")[0] < g.debug_get("pose_dim")[0]) == bool(chain)
    assert (sg.iterations, sg.trials, sg.solver_failures) == (so.iterations, so.trials, 0)
    assert sg.chi2_final == pytest.approx(so.chi2_final, rel=1e-8)
    assert max(_pose_delta(g.get_keyframes(), o.get_keyframes(), pkg)) < 1e-8
    g.close(); o.close()


def test_save_restore_replays_identically(pkg, hip):
    w = pkg.window.make_window(8, 200, 40, imu=True, seed=107)
    g = pkg.new_problem(); g.upload_window(w)
    g.save_state()
    a = g.optimize(4); ka = g.get_keyframes()
    g.restore_state()
    b = g.optimize(4); kb = g.get_keyframes()
    assert a.chi2_final == pytest.approx(b.chi2_final, rel=1e-12)
    assert np.abs(ka["P"] - kb["P"]).max() < 1e-12
    g.close()


def test_abort_and_errors(pkg, hip):
    w = pkg.window.make_window(5, 40, 8, imu=True, seed=15)
    g = pkg.new_problem(); g.upload_window(w)
    st = g.optimize(5, np.ones(1, np.uint8))
    assert st.iterations == 0 and st.stop_reason == 2
    bad = w["po_pt"].copy(); bad[0], bad[-1] = bad[-1], bad[0]
    with pytest.raises(pkg.abi.PlbaError):
        g.set_point_obs(bad, w["po_kf"], w["po_uv"], w["po_w"])
    g.close()
    e = pkg.new_problem()
    with pytest.raises(pkg.abi.PlbaError):
        e.optimize(1)
    e.close()


def test_smoke_entry():
    import __graft_entry__ as ge
    ge.smoke()


def _marg_compare(pr, po, eps=1e-8):
    assert (pr["n"], pr["m"]) == (po["n"], po["m"])
    assert list(pr["vid"]) == list(po["vid"]) and list(pr["size"]) == list(po["size"]) and list(pr["idx"]) == list(po["idx"])
    sc = np.abs(po["Ar"]).max()
    assert np.abs(pr["Ar"] - po["Ar"]).max() < 1e-7 * sc
    assert np.abs(pr["br"] - po["br"]).max() < 1e-7 * max(np.abs(po["br"]).max(), 1.0)
    assert np.abs(pr["x0"] - po["x0"]).max() < 1e-12
    # the prior is defined through J0^T J0 and J0^T r0 (eigenvector order/sign is arbitrary, SURVEY B-Q6)
    w, V = np.linalg.eigh(po["Ar"])
    keep = w > eps
    Ath = (V[:, keep] * w[keep]) @ V[:, keep].T
    assert np.abs(pr["J0"].T @ pr["J0"] - Ath).max() < 1e-7 * sc
    assert np.abs(pr["J0"].T @ pr["r0"] - po["J0"].T @ po["r0"]).max() < 1e-6 * max(np.abs(po["br"]).max(), 1.0)
    # b'^T A'^+ b' weights each eigen-direction by 1/lambda: conditioned like A', so only a loose check.  (A constant offset of chi2 that
    # cancels in every LM decision.  Against the 40-digit evaluation fp64 reaches 1e-5 .. 1e-3, device and oracle alike — DESIGN.md 6 —
    # and two equally valid roundings of the optimisation before it (chain segments of 4 or of 8 blocks) moved the n = 132 window's
    # value from 3e-5 to 1.6e-3 of the oracle's: 3e-3 is that measured spread, not a target.)
    assert pr["r0"] @ pr["r0"] == pytest.approx(po["r0"] @ po["r0"], rel=3e-3)


def test_marginalization_parity(pkg, orc, hip):
    """K9 against the oracle's IMU/marginalization.cpp restatement, then the chained case where the
    old prior itself is a factor (mapHandler.cpp:6167-6188)."""
    w = pkg.window.make_window(12, 260, 50, imu=True, seed=21)
    g, o = _pair(pkg, orc, w)
    pkg.protocol.local_ba(g); pkg.protocol.local_ba(o)
    pg, po = g.marginalize(0, 50), o.marginalize(0, 50)
    _marg_compare(pg, po)
    g.close(); o.close()
    # chained: feed the oracle's prior to both sides, optimise, marginalize again
    w["prior"] = po
    g, o = _pair(pkg, orc, w)
    g.optimize(3); o.optimize(3)
    pg2, po2 = g.marginalize(0, 50), o.marginalize(0, 50)
    _marg_compare(pg2, po2)
    g.close(); o.close()


def _marg_exact_cases():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    if here not in sys.path:
        sys.path.insert(0, here)
    import marg_cases
    return marg_cases.CASES, marg_cases.case_window, np.load(os.path.join(here, "marg_exact.npz"))


def test_marginalization_pinv_of_the_whole_dropped_block(pkg, orc, hip):
    """VERDICT r02 item 1.  The reference thresholds the eigenvalues of the WHOLE dropped block Amm at 1e-8
    (IMU/marginalization.cpp:351-362).  Default options (marg_exact = 1): the device takes the cheap block-by-block form only when
    its certificate proves it identical, else the dense eigen-decomposition of Amm (one-sided Jacobi on the stacked Jacobian's
    columns).  Arbiter: tests/golden/marg_exact.npz — the step evaluated at 40 digits (make_marg_exact.py) — because where far
    landmarks put eigenvalues of Amm around the threshold the fp64 oracle is itself only good to ~1e-4 (forming Amm in fp64 moves
    them; tests/test_oracle_marg.py pins that)."""
    cases, case_window, gold = _marg_exact_cases()
    for name, spec in cases:
        w = case_window(pkg, spec)
        g, o = _pair(pkg, orc, w)
        pg, po = g.marginalize(0, 50), o.marginalize(0, 50)
        path = g.debug_get("marg_path")
        g.close(); o.close()
        Ar, br, r0r0 = gold[name + "_Ar"], gold[name + "_br"], float(gold[name + "_r0r0"][0])
        assert (pg["m"], pg["n"]) == tuple(int(x) for x in gold[name + "_dims"][:2]) and list(pg["vid"]) == list(gold[name + "_vid"])
        sc, sb = np.abs(Ar).max(), max(np.abs(br).max(), 1.0)
        dev_A, dev_b = np.abs(pg["Ar"] - Ar).max() / sc, np.abs(pg["br"] - br).max() / sb
        orc_A = np.abs(po["Ar"] - Ar).max() / sc
        assert dev_A < 1e-10 and dev_b < 1e-10, (name, dev_A, dev_b)
        # far landmarks: the dense path must have been taken (the block-wise form is off by up to 3e-2 there, see below)
        assert int(path[0]) == (1 if spec.get("far", 1.0) >= 1e2 else 0), (name, path)
        assert dev_A <= orc_A + 1e-12, (name, dev_A, orc_A)
        # b'^T A'^+ b' weights each eigen-direction by 1 / lambda down to 1e-8: conditioned like A', fp64 reaches 1e-5 .. 1e-3
        # (device AND oracle, measured against the 40-digit value) — which is why _marg_compare checks it only to 3e-3
        dev_r, orc_r = abs(pg["r0"] @ pg["r0"] - r0r0) / r0r0, abs(po["r0"] @ po["r0"] - r0r0) / r0r0
        assert dev_r < 5e-4 and dev_r < 4.0 * orc_r + 1e-4, (name, dev_r, orc_r)
        # the prior itself: J0^T J0 = A' restricted to its eigenvalues above eps
        wv, V = np.linalg.eigh(Ar)
        keep = wv > 1e-8
        assert np.abs(pg["J0"].T @ pg["J0"] - (V[:, keep] * wv[keep]) @ V[:, keep].T).max() < 1e-7 * sc, name


def test_marginalization_forced_paths(pkg, orc, hip):
    """marg_exact = 2 (always dense) and 0 (always block-wise) against the 40-digit values: the dense form is exact everywhere; the
    block-wise one only where every discarded direction is a block-local null space (far = 1), and is off by 1e-3 .. 3e-2 of
    max |A'| on the far-landmark windows — the deviation VERDICT r02 asked to remove from the default path."""
    cases, case_window, gold = _marg_exact_cases()
    for name, spec in cases:
        w = case_window(pkg, spec)
        dev = {}
        for mode in (0, 2):
            g = pkg.new_problem(marg_exact=mode); g.upload_window(w)
            pg = g.marginalize(0, 50)
            assert int(g.debug_get("marg_path")[0]) == (1 if mode == 2 else 0)
            g.close()
            dev[mode] = np.abs(pg["Ar"] - gold[name + "_Ar"]).max() / np.abs(gold[name + "_Ar"]).max()
        assert dev[2] < 1e-10, (name, dev)
        if spec.get("far", 1.0) >= 1e2:
            assert 1e-4 < dev[0] < 0.2, (name, dev)
        else:
            assert dev[0] < 1e-10, (name, dev)


@pytest.mark.parametrize("K,dt,n_expect", [(11, 0.1, 96), (12, 0.1, 105), (15, 0.08, 132), (21, 0.05, 186)])
def test_marginalization_kept_blocks_beyond_100_dims(pkg, orc, hip, K, dt, n_expect):
    """VERDICT r02 item 4 / ADVICE: tracks over the whole window put every keyframe among the kept parameters — n = 15 + 9 (K - 2):
    105 at the reference's 12-keyframe window (src/mapHandler.cpp:6109-6188).  n <= 100: A' and V in one workgroup's LDS;
    n <= 140: A' in LDS, V in global memory (n > 128: three pairs per half-wave); beyond: two-sided Jacobi, two launches per round.
    Each against the oracle, first slide and chained slide (the old prior as a factor)."""
    w = pkg.window.make_window(K, 300, 60, imu=True, seed=77, kf_dt=dt, track=(K, K))
    g, o = _pair(pkg, orc, w)
    g.optimize(2); o.optimize(2)
    pg, po = g.marginalize(0, 50), o.marginalize(0, 50)
    assert pg["n"] == n_expect
    _marg_compare(pg, po)
    g.close(); o.close()
    w["prior"] = po
    g, o = _pair(pkg, orc, w)
    g.optimize(2); o.optimize(2)
    pg2, po2 = g.marginalize(0, 50), o.marginalize(0, 50)
    assert pg2["n"] == n_expect
    _marg_compare(pg2, po2)
    g.close(); o.close()


def test_sliding_window_with_device_prior(pkg, orc, hip):
    """config 4 shape: BA -> device marginalization -> next BA carries the device-built prior;
    the same prior fed to the oracle must give the same poses."""
    w = pkg.window.make_window(12, 260, 50, imu=True, seed=23)
    g = pkg.new_problem(); g.upload_window(w)
    pkg.protocol.local_ba(g)
    pr = g.marginalize(0, 50)
    g.close()
    w2 = pkg.window.make_window(12, 260, 50, imu=True, seed=23)
    w2["prior"] = pr
    g, o = _pair(pkg, orc, w2)
    sg, so = g.optimize(5), o.optimize(5)
    assert sg.chi2_final == pytest.approx(so.chi2_final, rel=1e-7)
    assert max(_pose_delta(g.get_keyframes(), o.get_keyframes(), pkg)) < 1e-7
    g.close(); o.close()


def test_ba_call_time_does_not_depend_on_other_problems_alive(pkg, hip):
    """with a second problem alive in the process, a BA call used to stall ~20 ms in the first stream
    operations of prepare().  Cause (tools/debug_e2e3.py): results were copied into pageable caller memory with hipMemcpy, the
    runtime pinned those pages, and when the host allocator returned them to the OS the driver evicted the process's GPU queues;
    which call paid depended on the allocator's state.  Every host copy now goes through the library's pinned staging area
    (plba_problem.h: plba_d2h / plba_h2d).  Asserted: a whole reference-shaped call with another problem alive stays within 2x
    of the solo call (measured: 9.2 vs 7.7 ms at configs[2]; here on a smaller window)."""
    import time
    import torch
    w = pkg.window.make_config(3, scale=0.25)

    def call():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p = pkg.new_problem(); p.upload_window(w)
        pkg.protocol.local_ba(p); pkg.protocol.results(p); p.close()
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    call(); call()
    solo = min(call() for _ in range(4))
    other = pkg.new_problem(); other.upload_window(w); other.optimize(2)
    call()
    busy = [call() for _ in range(8)]
    other.close()
    # (the stall was systematic — every call paid it; a single slow call on a shared box is not it: at most one of eight may exceed the bound)
    assert sum(b >= 2.0 * solo + 2e-3 for b in busy) <= 1 and sorted(busy)[len(busy) // 2] < 1.5 * solo + 1e-3, (solo, busy)


@pytest.mark.parametrize("K", [26, 34, 50, 77, 128, 200])
def test_segment_length_choice_predicts_the_plan_that_is_built(pkg, hip, K):
    """prepare() chooses the chain elimination's segment length by the number of dependent launches the multi-chain factorisation will need
    (twin_launch_estimate: the plan builder's arithmetic restated for every candidate's dense layout and band).  The estimate for the chosen
    length must be what the plan that is then BUILT needs — the two pieces of arithmetic are separate code."""
    w = pkg.window.make_window(K, 12 * K, 3 * K, imu=True, seed=0xC0DE + K)
    g = pkg.new_problem(); g.upload_window(w); g.optimize(1)
    assert int(g.debug_get("twin")[0]) == 1
    assert int(g.debug_get("fact_launches_estimate")[0]) == int(g.debug_get("fact_launches")[0]) + 1
    g.close()


def test_a_problem_handle_reused_across_windows_matches_fresh_handles(pkg, hip):
    """prepare() bump-allocates its small device buffers from two blocks the problem keeps and starts them over with every upload
    (csrc/plba_problem.h, DevBatch): a handle that takes window A, then a larger window B with IMU edges and a prior-free chain path,
    then A again — with gating, a state save / restore and a read-back in between — must give, bit for bit, what a fresh handle gives for
    each of them (the reference builds a fresh optimizer per local BA, src/mapHandler.cpp:5799; a binding that keeps the handle must not
    see the previous window)."""
    wa = pkg.window.make_window(6, 120, 30, imu=False, seed=11)
    wb = pkg.window.make_window(12, 900, 200, imu=True, seed=12, kf_dt=0.1, track=(6, 12), revisit=0.2)
    wc = pkg.window.make_config(3, scale=0.5)      # 51 k observations: the fused landmark passes, multi-chain factorisation

    def run(p, w):
        p.upload_window(w)
        s1 = p.optimize(3); p.gate_outliers(pkg.window.CHI2_GATE)
        lev = np.zeros(len(w["po_pt"]), np.uint8); lev[:7] = 1; p.set_levels(pkg.abi.EDGE_POINT, lev)      # (levels the caller sets are the window's, not the handle's)
        p.save_state(); s2 = p.optimize(3)
        kf, pts, lns = p.get_keyframes(), p.get_points(), p.get_lines()
        return (s1.chi2_final, s2.chi2_final, s2.trials), kf, pts, lns

    fresh = {}
    for name, w in (("a", wa), ("b", wb), ("c", wc)):
        p = pkg.new_problem(); fresh[name] = run(p, w); p.close()
    p = pkg.new_problem()
    for name, w in (("a", wa), ("c", wc), ("b", wb), ("b", wb), ("a", wa), ("b", wb), ("c", wc)):
        st, kf, pts, lns = run(p, w)
        fst, fkf, fpts, flns = fresh[name]
        assert st == fst, (name, st, fst)
        for k in fkf: assert np.array_equal(np.asarray(kf[k]), np.asarray(fkf[k])), (name, k)
        assert np.array_equal(pts, fpts) and np.array_equal(lns, flns), name
    p.close()


def test_edges_of_a_previous_window_are_refused_not_dereferenced(pkg, hip):
    """a re-used handle whose IMU edges still name keyframes 0 .. 11 of the previous window, given a 6-keyframe window through the raw
    setters (NOT upload_window, which clears them): plba_optimize must fail with PLBA_ERR_INVALID — before round 4's check the edges were
    uploaded as they were and the kernels read keyframes that do not exist"""
    wb = pkg.window.make_window(12, 300, 60, imu=True, seed=12)
    wa = pkg.window.make_window(6, 120, 30, imu=False, seed=11)
    p = pkg.new_problem(); p.upload_window(wb); p.optimize(1)
    k = wa["kf"]
    p.set_keyframes(k["vid_pvr"], k["vid_bias"], k["P"], k["V"], k["q"], k["bg"], k["ba"], k["dbg"], k["dba"], k["fixed_pvr"], k["fixed_bias"])
    p.set_points(wa["points"], wa.get("point_fixed")); p.set_lines(wa["lines"], wa.get("line_fixed"))
    p.set_point_obs(wa["po_pt"], wa["po_kf"], wa["po_uv"], wa["po_w"]); p.set_line_obs(wa["lo_ln"], wa["lo_kf"], wa["lo_l"], wa["lo_w"])
    with pytest.raises(Exception) as ei:
        p.optimize(1)
    assert "imu edge" in str(ei.value)
    p.set_imu_edges(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 142)), np.zeros((0, 81)), np.zeros((0, 36)))
    st = p.optimize(2)
    q = pkg.new_problem(); q.upload_window(wa); sq = q.optimize(2)
    assert st.chi2_final == sq.chi2_final
    p.close(); q.close()


def test_two_host_threads_one_problem_each_match_the_serial_runs(pkg, hip):
    """include/plba.h: "thread-compatible: one thread per plba_problem".  Two host threads, one problem each, their upload / optimize / gate
    / optimize calls interleaved by a barrier per step — the structure builds of both then run on the ONE host worker pool at the same time
    (HostPool's job serialisation and its asynchronous table fill) and their launches share the library's stream — must give,
    bit for bit, what each window gives when it runs alone."""
    import threading
    wa = pkg.window.make_config(3, scale=0.5)      # 51 k observations: fused landmark passes, asynchronous table fill on the pool
    wb = pkg.window.make_window(12, 900, 200, imu=True, seed=12, kf_dt=0.1, track=(6, 12), revisit=0.2)      # record-based passes: pair lists on the pool

    def run(w, step=None):
        p = pkg.new_problem()
        sync = step if step is not None else (lambda: None)
        sync(); p.upload_window(w)
        sync(); s1 = p.optimize(3)
        sync(); p.gate_outliers(pkg.window.CHI2_GATE)
        sync(); s2 = p.optimize(4)
        sync(); out = (s1.chi2_final, s2.chi2_final, s1.trials + s2.trials, p.get_keyframes(), p.get_points(), p.get_lines())
        p.close()
        return out

    serial = [run(wa), run(wb)]
    for rep in range(3):
        bar = threading.Barrier(2)
        res, err = [None, None], []

        def worker(i, w):
            try:
                res[i] = run(w, step=lambda: bar.wait(timeout=120))
            except Exception as e:      # noqa: BLE001
                err.append(e); bar.abort()
        th = [threading.Thread(target=worker, args=(i, w)) for i, w in enumerate((wa, wb))]
        for t in th: t.start()
        for t in th: t.join()
        assert not err, err
        for i in range(2):
            assert res[i][:3] == serial[i][:3], (rep, i, res[i][:3], serial[i][:3])
            for k in serial[i][3]: assert np.array_equal(res[i][3][k], serial[i][3][k]), (rep, i, k)
            assert np.array_equal(res[i][4], serial[i][4]) and np.array_equal(res[i][5], serial[i][5]), (rep, i)


def test_line_observations_keep_the_point_levels(pkg, hip):
    """set_point_obs -> set_levels(POINT) -> set_line_obs used to reset the point levels just set."""
    w = pkg.window.make_window(6, 120, 30, imu=False, seed=11)
    p = pkg.new_problem(); p.upload_window(w)
    lev = np.zeros(len(w["po_pt"]), np.uint8); lev[3:9] = 1
    p.set_point_obs(w["po_pt"], w["po_kf"], w["po_uv"], w["po_w"])
    p.set_levels(pkg.abi.EDGE_POINT, lev)
    p.set_line_obs(w["lo_ln"], w["lo_kf"], w["lo_l"], w["lo_w"])
    assert np.array_equal(p.get_levels(pkg.abi.EDGE_POINT), lev) and not p.get_levels(pkg.abi.EDGE_LINE).any()
    p.set_point_obs(w["po_pt"], w["po_kf"], w["po_uv"], w["po_w"])      # new point edges: level 0 again
    assert not p.get_levels(pkg.abi.EDGE_POINT).any()
    p.close()


def test_dense_solve_product_entry(pkg, hip):
    """plba_dense_solve (the facade's device solve for host-evaluated graphs beyond 384 dims) = the entry the tests know as plba_debug_dense_solve"""
    rng = np.random.default_rng(5)
    n = 400
    B = rng.standard_normal((n, n)); A = B @ B.T + n * np.eye(n); b = rng.standard_normal(n)
    p = pkg.new_problem()
    x, ok = p.dense_solve(A, b)
    x2, ok2 = p.debug_dense_solve(A, b)
    p.close()
    assert ok and ok2 and np.array_equal(x, x2)
    assert np.abs(x - np.linalg.solve(A, b)).max() < 1e-10 * max(np.abs(x).max(), 1.0)


def test_marg_eps_is_settable_like_the_reference_member(pkg, orc, hip):
    """MarginalizationInfo::eps is a public mutable member (IMU/marginalization.h:99); plba_set_marg_eps carries a caller's value into
    the thresholds of IMU/marginalization.cpp:353,365-366.  A threshold of 1e3 discards every kept direction below it: compared with
    the oracle at the same setting (and it must differ from the default's result)."""
    w = pkg.window.make_window(12, 260, 50, imu=True, seed=21)
    g, o = _pair(pkg, orc, w)
    pkg.protocol.local_ba(g); pkg.protocol.local_ba(o)
    p_def = g.marginalize(0, 50)
    g.set_marg_eps(1e3); o.set_marg_eps(1e3)
    pg, po = g.marginalize(0, 50), o.marginalize(0, 50)
    _marg_compare(pg, po, eps=1e3)
    assert np.abs(pg["J0"].T @ pg["J0"] - p_def["J0"].T @ p_def["J0"]).max() > 1.0      # directions with 1e-8 < lambda <= 1e3 are gone
    with pytest.raises(Exception):
        g.set_marg_eps(float("nan"))
    g.close(); o.close()


@pytest.mark.parametrize("K,dt", [(8, 0.1), (12, 0.1)])
def test_marginalization_prerotation_and_cold_jacobi_agree(pkg, orc, hip, K, dt):
    """Round 5: the Jacobi of the kept block starts from a Householder-tridiagonal + implicit-QL pre-rotation (one sweep over the
    numerically small directions instead of ~13 sweeps over everything); `diag` bit 3 keeps the cold start.  Same criteria, same accuracy
    class: both against the oracle, and against each other through what the prior is defined by (J0^T J0, J0^T r0; the eigenvector basis
    of a degenerate subspace is free).  Replaces Eigen::SelfAdjointEigenSolver at IMU/marginalization.cpp:364."""
    w = pkg.window.make_window(K, 300, 60, imu=True, seed=91, kf_dt=dt, track=(K, K))
    out = {}
    for diag in (0, 8):
        g = pkg.new_problem(diag=diag); g.upload_window(w)
        g.optimize(2)
        out[diag] = g.marginalize(0, 50)
        g.close()
    o = orc.new_problem(); o.upload_window(w); o.optimize(2)
    po = o.marginalize(0, 50); o.close()
    for diag in (0, 8):
        _marg_compare(out[diag], po)
    a, b = out[0], out[8]
    sc = np.abs(po["Ar"]).max()
    assert np.array_equal(a["Ar"], b["Ar"]) and np.array_equal(a["br"], b["br"])      # (the block itself does not depend on the eigen-solver)
    assert np.abs(a["J0"].T @ a["J0"] - b["J0"].T @ b["J0"]).max() < 1e-9 * sc
    assert np.abs(a["J0"].T @ a["r0"] - b["J0"].T @ b["r0"]).max() < 1e-7 * max(np.abs(po["br"]).max(), 1.0)
