"""The native / OpenMP build of the oracle (oracle/Makefile, -DPLBA_ORACLE_FAST: envelope Cholesky of the reduced system, landmark
loops over OpenMP threads with band-limited private accumulators) is only ever bench.py's cpu_baseline — but a baseline that computes
something else would be worthless, so it is held to the faithful single-thread build here: same LM decisions, same estimates up to
summation order.  (The algorithm both restate: g2o's Levenberg loop with the Schur complement, SURVEY App. A.)"""
import numpy as np
import pytest


def _run(prob, w, iters=(3, 4)):
    prob.upload_window(w)
    prob.optimize(iters[0])
    prob.gate_outliers()
    prob.optimize(iters[1])
    out = prob.get_keyframes(), prob.get_points(), prob.get_lines(), prob.trace()
    prob.close()
    return out


@pytest.mark.parametrize("threads", [1, 3])
@pytest.mark.parametrize("shape", ["imu", "visual", "prior_fixed"])
def test_fast_build_agrees_with_the_faithful_oracle(pkg, orc, shape, threads):
    if shape == "imu":
        w = pkg.window.make_window(10, 240, 50, imu=True, seed=4242)
    elif shape == "visual":
        w = pkg.window.make_window(7, 150, 40, imu=False, seed=4243)
    else:
        w = pkg.window.make_window(9, 200, 40, imu=True, seed=4244)
        w["point_fixed"] = (np.arange(200) % 17 == 0).astype(np.uint8)
        w["kf_fixed"] = np.zeros(9, np.uint8); w["kf_fixed"][0] = 1
    kf_a, pt_a, ln_a, tr_a = _run(orc.new_problem(), w)
    kf_b, pt_b, ln_b, tr_b = _run(orc.new_fast_problem(threads=threads), w)
    assert len(tr_a) == len(tr_b) and len(tr_a) >= 4      # (the trace of the last optimize call)
    for a, b in zip(tr_a, tr_b):
        assert (a["iteration"], a["trial"], a["accepted"], a["solver_ok"]) == (b["iteration"], b["trial"], b["accepted"], b["solver_ok"])
        assert abs(a["chi2_trial"] - b["chi2_trial"]) <= 1e-8 * max(abs(a["chi2_trial"]), 1.0)
        assert abs(a["lam"] - b["lam"]) <= 1e-8 * abs(a["lam"])
    for k in kf_a:
        assert np.abs(kf_a[k] - kf_b[k]).max() < 1e-8, k
    assert np.abs(pt_a - pt_b).max() < 1e-7 and (ln_a.size == 0 or np.abs(ln_a - ln_b).max() < 1e-7)
