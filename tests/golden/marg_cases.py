"""The windows of tests/golden/marg_exact.npz (shared by make_marg_exact.py, which writes the file, and the tests that read it)."""
import numpy as np

CASES = [("far1", dict(far=1.0)), ("far1e2", dict(far=1e2)), ("far1e3", dict(far=1e3)), ("far1e6", dict(far=1e6)), ("k12full", dict(full=12))]


def case_window(pkg, spec):
    """far: the landmarks the oldest keyframe sees, pushed out along their rays — Jacobians ~ fx / depth, so the information a view
    gives on a landmark falls like 1 / depth^2 and its depth direction like 1 / depth^4, through the 1e-8 threshold.
    full: a K-keyframe window whose tracks span the whole window (every keyframe ends up among the kept parameters)."""
    if "far" in spec:
        far = spec["far"]
        w = pkg.window.make_window(6, 120, 20, imu=True, seed=31, outlier_frac=0.0)
        P0 = w["kf"]["P"][0]
        sp = sorted(set(w["po_pt"][w["po_kf"] == 0].tolist())); sl = sorted(set(w["lo_ln"][w["lo_kf"] == 0].tolist()))
        w["points"][sp] = P0 + far * (w["points"][sp] - P0)
        w["lines"][sl] = np.tile(P0, 2) + far * (w["lines"][sl] - np.tile(P0, 2))
        return w
    K = spec["full"]
    return pkg.window.make_window(K, 300, 60, imu=True, seed=77, kf_dt=0.1, track=(K, K))
