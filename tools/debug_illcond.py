import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
from oracle import oracle as orc
pkg = ge.load_package()
w = pkg.window.make_window(5, 60, 10, imu=False, seed=13)
w["points"] = w["points"] + np.random.default_rng(0).normal(size=w["points"].shape) * 1.5
for lam in (1e-3, 1.0, 1e3):
    g = pkg.new_problem(); g.upload_window(w); o = orc.new_problem(); o.upload_window(w)
    g.debug_build(lam, False); o.debug_build(lam, False)
    Hg, Ho = g.debug_get("Hschur"), o.debug_get("Hschur"); P = int(g.debug_get("pose_dim")[0])
    Hg = Hg.reshape(P, P); Ho = Ho.reshape(P, P); bg, bo = g.debug_get("bschur"), o.debug_get("bschur")
    print("lam", lam, "Hschur rel diff", np.abs(Hg - Ho).max() / np.abs(Ho).max(), "b rel diff", np.abs(bg - bo).max() / np.abs(bo).max(), "cond", np.linalg.cond(Ho))
    g.debug_build(lam, True); o.debug_build(lam, True)
    xg, xo = g.debug_get("x")[:P], o.debug_get("x")[:P]
    xl = np.linalg.solve(Ho.astype(np.longdouble).astype(np.float64), bo)
    import scipy.linalg as sl
    xr = sl.solve(Ho, bo, assume_a='pos')
    print("   |xg-xo|/|xo|", np.linalg.norm(xg - xo) / np.linalg.norm(xo), "|xg-xr|", np.linalg.norm(xg - xr) / np.linalg.norm(xr), "|xo-xr|", np.linalg.norm(xo - xr) / np.linalg.norm(xr),
          "res g", np.linalg.norm(Ho @ xg - bo) / np.linalg.norm(bo), "res o", np.linalg.norm(Ho @ xo - bo) / np.linalg.norm(bo))
    g.close(); o.close()
