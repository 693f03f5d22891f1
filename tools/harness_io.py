"""Build and feed tools/localba_harness.cpp (the reference's local-BA call sites re-enacted on the g2o facade): shared by
tests/test_facade.py and bench.py's `facade_ba_call_ms` leg."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "_build_localba_harness")


def build_harness():
    import __graft_entry__ as g
    g.build_hip()
    src = os.path.join(ROOT, "tools", "localba_harness.cpp")
    deps = [src, os.path.join(ROOT, "include", "plba_g2o", "g2o_compat.h"), os.path.join(ROOT, "include", "plba_g2o", "vio_init.h"), os.path.join(ROOT, "include", "plba.h")]
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wno-unknown-pragmas", "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(ROOT, "pl-inertial-slam_amd", "csrc"), src, "-o", EXE,
                               "-L", os.path.join(ROOT, "pl-inertial-slam_amd"), "-lplba_hip",
                               "-Wl,-rpath," + os.path.join(ROOT, "pl-inertial-slam_amd")])
    return EXE


def write_window(w, path, do_marg=1, max_kf=12):
    K, Np, Nl = len(w["kf"]["P"]), len(w["points"]), len(w["lines"])
    Ep, El = len(w["po_pt"]), len(w["lo_ln"])
    im = w["imu"]
    M = len(im["kf_i"])
    with open(path, "wb") as f:
        np.array([K, Np, Nl, Ep, El, M, do_marg, max_kf], np.int32).tofile(f)
        c = w["cam"]
        np.array([c["fx"], c["fy"], c["cx"], c["cy"]], np.float64).tofile(f)
        np.asarray(c["Rbc"], np.float64).ravel().tofile(f); np.asarray(c["Pbc"], np.float64).tofile(f)
        np.asarray(w["gw"], np.float64).tofile(f)
        np.array([w["huber"][k] for k in range(4)], np.float64).tofile(f)
        (w["kf"]["vid_pvr"] // 2).astype(np.int32).tofile(f)
        for k in ("P", "V", "q", "bg", "ba"):
            np.ascontiguousarray(w["kf"][k], np.float64).tofile(f)
        np.ascontiguousarray(w["points"], np.float64).tofile(f); np.ascontiguousarray(w["lines"], np.float64).tofile(f)
        w["po_pt"].astype(np.int32).tofile(f); w["po_kf"].astype(np.int32).tofile(f)
        np.ascontiguousarray(w["po_uv"], np.float64).tofile(f); (1.0 / w["po_w"]).astype(np.float64).tofile(f)
        w["lo_ln"].astype(np.int32).tofile(f); w["lo_kf"].astype(np.int32).tofile(f)
        np.ascontiguousarray(w["lo_l"], np.float64).tofile(f); (1.0 / w["lo_w"]).astype(np.float64).tofile(f)
        np.ascontiguousarray(im["preint"], np.float64).tofile(f); np.ascontiguousarray(im["info_pvr"], np.float64).tofile(f)
        np.ascontiguousarray(im["info_bias"], np.float64).tofile(f)
