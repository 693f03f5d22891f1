"""The fused landmark-major passes (k_lm_schur / k_lm_gather / k_lm_trial, csrc/plba_lm_dev.h) THROUGH REJECTED TRIALS at the sizes where
they are the product's default path (>= 40 k observations), with DEFAULT options, against the exact trajectory: the quad-precision build
of the oracle (tests/golden/fused_overshoot_quad.json, generator make_fused_overshoot_quad.py; that build is itself pinned to the
40-digit third implementation by tests/test_lm_trace.py).  Replaces g2o's push / solve / update / pop of a rejected step
 around src/mapHandler.cpp:6038-6069.

Per case: the LM decisions (iteration, trial, accepted, solver_ok) must be the quad run's; lambda / chi2 of every trial and the final
keyframe states must be no further from the quad run than 4 x the fp64 ORACLE is (floored at 1e-9: where the oracle is at rounding level
so must the device be) — the device is not allowed to be a worse fp64 implementation of the path than the CPU one.
  rej40k_c   3 rejected trials, well conditioned: oracle64 - quad = 1e-13  -> the device is held to 1e-9
  rej40k_a   6 rejected trials (one iteration rejects five times): oracle64 - quad = 1.5e-4
  rej40k_b   started at lambda = 1 against IMU information of 1e10: the fp64 ORACLE leaves the exact trajectory (7 rejections where the
             exact run has none, 8e-2 apart at the end).  Arbitrated: the device must stay within 4 x the oracle's distance; its own
             decisions and distance are printed (-s) and recorded in DESIGN.md
  rejected_small   test_gpu_parity.py::test_rejected_trials_with_imu_edges' window on the fused passes (lm_fused = 2), 12 rejected trials"""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(pkg, name):
    if GOLD not in sys.path:
        sys.path.insert(0, GOLD)
    import fused_overshoot_cases as fc
    with open(os.path.join(GOLD, "fused_overshoot_quad.json")) as f:
        fx = json.load(f)[name]
    w = fc.window(pkg, name)
    assert (int(w["meta"]["Ep"]), int(w["meta"]["El"])) == (fx["meta"]["Ep"], fx["meta"]["El"])
    return w, fx, fc.CASES[name]


def _run(pkg, w, c, **opts):
    g = pkg.new_problem(user_lambda_init=c["lambda_init"], **opts)
    g.upload_window(w)
    st = g.optimize(c["iters"])
    out = (st, g.trace(), g.get_keyframes(), int(g.debug_get("lm_fused")[0]))
    g.close()
    return out


def _decisions(tr):
    return [(t["iteration"], t["trial"], t["accepted"], t["solver_ok"]) for t in tr]


def _dist(kf, ref):
    return max(float(np.abs(np.asarray(kf[k]) - np.asarray(ref[k])).max()) for k in ref)


def _trace_dist(tr, ref):
    """largest relative distance of lambda / chi2 over the trials both runs have"""
    d = 0.0
    for a, b in zip(tr, ref):
        for k in ("lam", "chi2_current", "chi2_trial"):
            d = max(d, abs(a[k] - b[k]) / max(abs(b[k]), 1e-300))
    return d


# measured on MI355X (round 4), |HIP - quad| on the final keyframe states / on lambda, chi2 over the trace — next to the fp64 oracle's:
#   rej40k_c 1.1e-14 / 2.5e-13 (oracle64 1.1e-13 / 3.0e-12);  rej40k_a 1.0e-11 / 1.0e-8 (oracle64 1.5e-4 / 2.2e-2);
#   rejected_small 2.2e-10 / 1.3e-6 (oracle64 4.0e-4 / 0.63).  The square-root form of the fused Schur pass (A' = Hpl R^-T, S -= A' A'^T)
#   does not cancel where Hpp - Hpl D Hpl^T does.  BOUND = those x ~100, so that a regression towards the oracle's accuracy fails.
BOUND = {"rej40k_c": (1e-9, 1e-9), "rej40k_a": (1e-9, 1e-6), "rejected_small": (1e-7, 1e-4)}


@pytest.mark.parametrize("name", ["rej40k_c", "rej40k_a", "rejected_small"])
def test_fused_passes_follow_the_exact_trajectory_through_rejected_trials(pkg, hip, name):
    w, fx, c = _load(pkg, name)
    st, tr, kf, fused = _run(pkg, w, c, **({"lm_fused": 2} if name == "rejected_small" else {}))      # DEFAULT options at >= 40 k observations
    assert fused == 1, "the fused landmark passes did not run"
    q, o = fx["trace"], fx["trace_fp64_oracle"]
    assert sum(1 - t["accepted"] for t in q) >= 3
    assert _decisions(tr) == _decisions(q)
    d_or, d_hip = max(fx["fp64_oracle_abs_diff"].values()), _dist(kf, fx["kf"])
    t_or, t_hip = _trace_dist(o, q), _trace_dist(tr, q)
    print("%s: |HIP - quad| %.2e (oracle64 %.2e) states, %.2e (oracle64 %.2e) trace" % (name, d_hip, d_or, t_hip, t_or))
    assert d_hip <= max(4 * d_or, 1e-9) and t_hip <= max(4 * t_or, 1e-9)
    assert d_hip <= BOUND[name][0] and t_hip <= BOUND[name][1]
    assert st.chi2_final == pytest.approx(fx["chi2_final"], rel=max(4 * t_or, 1e-9))


def test_where_fp64_solvers_part_the_device_is_no_further_from_exact_than_the_oracle(pkg, hip):
    w, fx, c = _load(pkg, "rej40k_b")
    st, tr, kf, fused = _run(pkg, w, c)
    assert fused == 1
    q, o = fx["trace"], fx["trace_fp64_oracle"]
    assert _decisions(o) != _decisions(q), "the case is meant to be one where the fp64 oracle leaves the exact trajectory"
    d_or, d_hip = max(fx["fp64_oracle_abs_diff"].values()), _dist(kf, fx["kf"])
    n_same = next((i for i, (a, b) in enumerate(zip(_decisions(tr), _decisions(q))) if a != b), min(len(tr), len(q)))
    n_same_or = next((i for i, (a, b) in enumerate(zip(_decisions(o), _decisions(q))) if a != b), min(len(o), len(q)))
    print("rej40k_b: |HIP - quad| %.2e, |oracle64 - quad| %.2e; trials in step with the exact run: HIP %d of %d, oracle64 %d" % (d_hip, d_or, n_same, len(q), n_same_or))
    assert d_hip <= 4 * d_or
    assert n_same >= n_same_or      # it follows the exact decisions at least as long as the CPU fp64 implementation does
    # measured (round 4): the device stays in step with the exact run through all 8 trials and ends 3.9e-7 from it; the fp64 oracle
    # leaves it at the sixth trial and ends 8.2e-2 away
    assert n_same == len(q) and d_hip <= 1e-4
    # the first damped solve (identical inputs): the device's chi2 must be at least as close to the exact one as the oracle's
    assert abs(tr[0]["chi2_trial"] - q[0]["chi2_trial"]) <= 4 * abs(o[0]["chi2_trial"] - q[0]["chi2_trial"]) + 1e-9 * q[0]["chi2_trial"]


def test_record_based_passes_on_the_same_windows(pkg, hip):
    """the record-based passes (lm_fused = 0) on the well-conditioned case: same bar — both landmark paths are the product somewhere"""
    w, fx, c = _load(pkg, "rej40k_c")
    st, tr, kf, fused = _run(pkg, w, c, lm_fused=0)
    assert fused == 0
    assert _decisions(tr) == _decisions(fx["trace"])
    assert _dist(kf, fx["kf"]) <= 1e-9 and _trace_dist(tr, fx["trace"]) <= 1e-9
