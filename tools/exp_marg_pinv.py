"""Experiment: a window in which the thresholded (eigenvalue <= marg_eps) directions of the dropped block Amm are NOT block-local:
the oldest keyframe is tied to the rest of the window only through its landmarks, whose other observations carry (almost) no
information.  Prints the deviation of the device's structured pseudo-inverse from the oracle's dense one."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import __graft_entry__ as g
from oracle import oracle as orc

pkg = g.load_package()
for far in (1.0, 1e3, 3e3, 1e4, 3e4, 1e5, 1e6, 1e7):
    w = pkg.window.make_window(6, 120, 20, imu=True, seed=31, outlier_frac=0.0)
    P0 = w["kf"]["P"][0]
    seen0 = sorted(set(w["po_pt"][w["po_kf"] == 0].tolist()))
    w["points"][seen0] = P0 + far * (w["points"][seen0] - P0)            # Jacobians ~ fx / depth: information ~ 1.6e5 / depth^2 per view
    seenl = sorted(set(w["lo_ln"][w["lo_kf"] == 0].tolist()))
    w["lines"][seenl] = np.tile(P0, 2) + far * (w["lines"][seenl] - np.tile(P0, 2))
    a = pkg.new_problem(); a.upload_window(w); b = orc.new_problem(); b.upload_window(w)
    pa, pb = a.marginalize(0, 50), b.marginalize(0, 50)
    sc = np.abs(pb["Ar"]).max()
    dA = np.abs(pa["Ar"] - pb["Ar"])
    print("depth x%.0e: n=%d  max|Ar_dev - Ar_orc| = %.3e (abs) = %.3e x max|Ar_orc|;  |br| dev %.3e (abs), max|br| %.3e" %
          (far, pb["n"], dA.max(), dA.max() / sc, np.abs(pa["br"] - pb["br"]).max(), np.abs(pb["br"]).max()))
    a.close(); b.close()
