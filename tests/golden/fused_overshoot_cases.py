"""Windows on which damped steps are REJECTED, at the sizes where the fused landmark-major passes are the product's default (>= 40 k
observations) — shared by the generator of their quad-precision fixture (make_fused_overshoot_quad.py) and tests/test_fused_overshoot.py.

30 keyframes / 8000 points / 1600 lines / IMU (41 k observations) as generated, landmarks then perturbed by `lm_sigma` metres
and the run started at a damping `lambda_init` that is too small for that start: Gauss-Newton-like steps on a problem that is not yet in
its quadratic basin overshoot and are rejected (rho < 0 with a successful factorisation — NOT solver failures).  The three recipes were
picked (with the fp64 oracle, tools-free: a loop over seeds / sigmas / dampings) for: >= 3 rejected trials, every factorisation
successful, and every |rho| >= 0.1 — so that an accept / reject decision is a property of the problem, not of the last bits of a solver.
The recipe of overshoot_cases.py (rotations off by 0.6 rad) does NOT scale to this size: at 30 keyframes the fp64 oracle's Schur
complement loses positive definiteness there (five factorisations fail where the quad build succeeds) and no two solvers share a
trajectory.

  rejected_small  test_gpu_parity.py::test_rejected_trials_with_imu_edges' own 10-keyframe window (lambda_init = 1e-6, landmarks perturbed
                  by a metre; run with lm_fused = 2 on the device): the case the forced-fused suite of round 3 failed on
"""
import numpy as np

CASES = {
    "rej40k_a": dict(seed=0x5EED42, K=30, Np=8000, Nl=1600, lm_sigma=0.6, lambda_init=100.0, iters=8),
    "rej40k_b": dict(seed=0x5EED41, K=30, Np=8000, Nl=1600, lm_sigma=0.6, lambda_init=1.0, iters=8),
    "rej40k_c": dict(seed=0x5EED41, K=30, Np=8000, Nl=1600, lm_sigma=0.3, lambda_init=100.0, iters=8),
    "rejected_small": dict(seed=77, K=10, Np=150, Nl=30, lm_sigma=1.0, lambda_init=1e-6, iters=6, points_only=True),
}


def window(pkg, name):
    c = CASES[name]
    w = pkg.window.make_window(c["K"], c["Np"], c["Nl"], imu=True, seed=c["seed"])
    rng = np.random.default_rng(1)
    w["points"] = w["points"] + rng.normal(size=w["points"].shape) * c["lm_sigma"]
    if not c.get("points_only"):
        w["lines"] = w["lines"] + rng.normal(size=w["lines"].shape) * c["lm_sigma"]
    return w
