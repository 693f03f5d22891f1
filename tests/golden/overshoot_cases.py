"""The overshooting IMU windows of tests/test_gpu_parity.py, shared with the generator of their quad-precision fixture
(make_overshoot_quad.py)."""
import numpy as np

CASES = [("seed81", dict(seed=81)), ("seed77", dict(seed=77))]
LAMBDA_INIT, ITERS = 1e4, 8


def overshoot_window(pkg, seed, rot=0.6, vel=20.0, pts=0.1, K=10, Np=150, Nl=30):
    """an IMU window started so far from the optimum that damped Gauss-Newton steps overshoot and get rejected: rotations off
    by `rot` rad, velocities by `vel` m/s, points by `pts` m (small: a point pushed through the camera plane makes the Schur
    complement cancel catastrophically, and then NO two fp64 solvers agree)"""
    W = pkg.window
    w = W.make_window(K, Np, Nl, imu=True, seed=seed)
    rng = np.random.default_rng(seed)
    kf = w["kf"]
    q = kf["q"].copy()
    for k in range(1, K):
        q[k] = W.quat_from_R(W.R_from_quat(q[k]) @ W.exp_so3(rng.normal(size=3) * rot))
    kf["q"] = q
    kf["V"] = kf["V"] + np.vstack([np.zeros((1, 3)), rng.normal(size=(K - 1, 3)) * vel])
    w["points"] = w["points"] + rng.normal(size=w["points"].shape) * pts
    return w
