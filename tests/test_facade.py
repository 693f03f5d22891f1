"""Boundary 1: code written against the reference's g2o API (IMU/g2otypes.h, IMU/marginalization.h, g2o core
headers) runs on the HIP path.  tools/localba_harness.cpp re-enacts the call-site protocol of
MapHandler::localBundleAdjustmentWithImuAndMarg (src/mapHandler.cpp:5741-6254) through include/plba_g2o/ and
include/g2o/; its results must equal the C-ABI path and the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from harness_io import EXE, build_harness, write_window  # noqa: E402


def read_result(path, K, Np, Nl):
    with open(path, "rb") as f:
        gp, gl, n, m, nv = np.fromfile(f, np.int32, 5)
        chi = np.fromfile(f, np.float64, 1)[0]
        r = dict(gated=(int(gp), int(gl)), chi2=chi, n=int(n), m=int(m))
        r["P"] = np.fromfile(f, np.float64, 3 * K).reshape(K, 3); r["V"] = np.fromfile(f, np.float64, 3 * K).reshape(K, 3)
        r["q"] = np.fromfile(f, np.float64, 4 * K).reshape(K, 4)
        r["dbg"] = np.fromfile(f, np.float64, 3 * K).reshape(K, 3); r["dba"] = np.fromfile(f, np.float64, 3 * K).reshape(K, 3)
        r["points"] = np.fromfile(f, np.float64, 3 * Np).reshape(Np, 3); r["lines"] = np.fromfile(f, np.float64, 6 * Nl).reshape(Nl, 6)
        r["vid"] = np.fromfile(f, np.int32, nv); r["size"] = np.fromfile(f, np.int32, nv); r["idx"] = np.fromfile(f, np.int32, nv)
        r["J0"] = np.fromfile(f, np.float64, n * n).reshape(n, n).T.copy(); r["r0"] = np.fromfile(f, np.float64, n)
    return r


def test_harness_compiles_against_the_facade_headers():
    """Source compatibility: the reference-shaped call site builds with g++ against include/g2o + include/plba_g2o
    and links to libplba_hip.so (no GPU needed to build)."""
    exe = build_harness()
    assert os.access(exe, os.X_OK)
    out = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
    for sym in ("plba_create", "plba_optimize", "plba_set_point_obs", "plba_marginalize_factors", "plba_get_edge_chi2"):
        assert sym in out


@pytest.mark.gpu
def test_call_site_protocol_through_the_facade(pkg, orc, hip, tmp_path):
    w = pkg.window.make_window(12, 260, 50, imu=True, seed=31)
    exe = build_harness()
    win, res = str(tmp_path / "w.bin"), str(tmp_path / "r.bin")
    write_window(w, win)
    subprocess.check_call([exe, win, res], timeout=120)
    r = read_result(res, 12, len(w["points"]), len(w["lines"]))
    # the same protocol through the C ABI (python) and through the oracle
    g = pkg.new_problem(); g.upload_window(w)
    out = pkg.protocol.local_ba(g, marginalize=True)
    ref = pkg.protocol.results(g)
    assert r["gated"] == out["gated"]
    for k in ("P", "V", "q", "dbg", "dba", "points", "lines"):
        assert np.abs(r[k] - ref[k]).max() < 1e-9, k
    pr = out["prior"]
    assert r["n"] == pr["n"] and r["m"] == pr["m"] and list(r["vid"]) == list(pr["vid"])
    assert list(r["idx"]) == list(pr["idx"] + pr["m"])            # the reference keeps idx including m (marginalization.h:85-87)
    sc = np.abs(pr["Ar"]).max()
    assert np.abs(r["J0"].T @ r["J0"] - pr["J0"].T @ pr["J0"]).max() < 1e-8 * sc
    assert np.abs(r["J0"].T @ r["r0"] - pr["J0"].T @ pr["r0"]).max() < 1e-7 * max(np.abs(pr["br"]).max(), 1)
    g.close()
    o = orc.new_problem(); o.upload_window(w)
    pkg.protocol.local_ba(o)
    ro = pkg.protocol.results(o)
    assert np.abs(r["P"] - ro["P"]).max() < 1e-5 and np.abs(r["q"] - ro["q"]).max() < 1e-5
    o.close()


@pytest.mark.gpu
def test_a_graph_built_in_another_order_gives_the_same_results(pkg, hip, tmp_path):
    """Round 5: the facade fills the C ABI's arrays while the call site inserts (flatten-at-insertion) — valid as long as the graph
    arrives in the call site's own shape.  `scrambled` builds the SAME graph the way g2o also accepts: landmark vertices in descending id
    order, every edge after every vertex, the landmarks' edges in descending order too, a measurement set again after addEdge.  The facade
    must notice, rebuild the arrays from the objects (sorting the observations landmark-major itself) and give the ordered build's
    results: same landmark indices (rank by id), same observation order within a landmark — bit for bit."""
    w = pkg.window.make_window(12, 260, 50, imu=True, seed=31)
    exe = build_harness()
    win = str(tmp_path / "w.bin")
    write_window(w, win)
    out = {}
    for mode in ("ordered", "scrambled", "late"):
        res = str(tmp_path / (mode + ".bin"))
        subprocess.check_call([exe, win, res] if mode == "ordered" else [exe, mode, win, res], timeout=120)
        out[mode] = read_result(res, 12, len(w["points"]), len(w["lines"]))
    # `late`: the ordered build, but every point vertex is inserted with a placeholder estimate and gets its real one (and a fixed flag
    # set and cleared again) after the last edge: the facade captures estimates at addVertex and must write the later ones through
    for other in ("scrambled", "late"):
        a, b = out["ordered"], out[other]
        assert a["gated"] == b["gated"] and a["chi2"] == b["chi2"], other
        for k in ("P", "V", "q", "dbg", "dba", "points", "lines"):
            assert np.array_equal(a[k], b[k]), (other, k)
    # (the prior is not compared: the call site's factor selection walks ITS edge list and stops after NUM + 1 — another list order, other factors)


def make_nomarg_window(pkg, seed=41, K=12, n_fixed=3):
    """The graph MapHandler::localBundleAdjustmentWithImu (USE_MARG off, src/mapHandler.cpp:5086-5739) builds: keyframes
    [0, n_fixed) are covisible keyframes OUTSIDE the sliding window (fixed PVR vertex, NO bias vertex, :5220-5231), keyframe
    n_fixed is the window's RefKeyframe predecessor (fixed PVR + bias, :5233-5240, source of the first IMU edge), the rest is
    the window (nothing fixed: firstFix is false when a predecessor exists, :5157-5166).  Landmarks are the `local` ones —
    observed by at least one window keyframe — with ALL their observations, also those in the fixed keyframes."""
    w = pkg.window.make_window(K, 300, 60, imu=True, seed=seed)
    role = np.zeros(K, np.int32); role[:n_fixed] = 1; role[n_fixed] = 2
    kf = w["kf"]
    kf["vid_bias"] = kf["vid_bias"].copy(); kf["vid_bias"][:n_fixed] = -1
    kf["fixed_pvr"] = (role != 0).astype(np.uint8); kf["fixed_bias"] = (role == 2).astype(np.uint8)

    def local(lm, kfs, N):
        keep = np.zeros(N, bool)
        np.logical_or.at(keep, lm, role[kfs] == 0)
        new = np.cumsum(keep) - 1
        sel = keep[lm]
        return keep, sel, new
    keep, sel, new = local(w["po_pt"], w["po_kf"], len(w["points"]))
    w["points"] = w["points"][keep]
    w["po_pt"], w["po_kf"], w["po_uv"], w["po_w"] = new[w["po_pt"][sel]].astype(np.int32), w["po_kf"][sel], w["po_uv"][sel], w["po_w"][sel]
    keep, sel, new = local(w["lo_ln"], w["lo_kf"], len(w["lines"]))
    w["lines"] = w["lines"][keep]
    w["lo_ln"], w["lo_kf"], w["lo_l"], w["lo_w"] = new[w["lo_ln"][sel]].astype(np.int32), w["lo_kf"][sel], w["lo_l"][sel], w["lo_w"][sel]
    im = dict(w["imu"]); m = im["kf_i"] >= n_fixed
    for k in ("kf_i", "kf_j", "preint", "info_pvr", "info_bias"):
        im[k] = im[k][m]
    w["imu"] = im
    w["role"] = role
    return w


@pytest.mark.gpu
def test_use_marg_off_shape_against_the_oracle(pkg, orc, hip):
    """VERDICT r01 missing #3 / next #8: fixed covisible keyframes WITHOUT bias vertex mixed with the IMU-chained window, a fixed
    RefKeyframe carrying PVR + bias and an IMU edge into the window, two-stage solve, then the culling decision — through the
    C ABI against the oracle"""
    w = make_nomarg_window(pkg)
    assert (w["kf"]["vid_bias"] < 0).sum() == 3 and w["role"][w["po_kf"]].max() >= 1      # observations in fixed keyframes exist
    g = pkg.new_problem(); g.upload_window(w)
    o = orc.new_problem(); o.upload_window(w)
    g.debug_build(7.0, False); o.debug_build(7.0, False)
    assert g.debug_get("pose_dim")[0] == o.debug_get("pose_dim")[0] == 8 * 15
    for name in ("err_pt", "err_ln", "err_pvr", "err_bias", "bp", "bschur", "Hschur", "chi2", "maxdiag"):
        a, b = g.debug_get(name), o.debug_get(name)
        assert np.abs(a - b).max() <= 1e-9 * max(np.abs(b).max(), 1e-300), name
    rg, ro = pkg.protocol.local_ba(g), pkg.protocol.local_ba(o)
    assert rg["gated"] == ro["gated"]
    assert (rg["stage2"].iterations, rg["stage2"].trials, rg["stage2"].solver_failures) == (ro["stage2"].iterations, ro["stage2"].trials, 0)
    kg, ko = g.get_keyframes(), o.get_keyframes()
    for k in ("P", "V", "q", "dbg", "dba"):
        assert np.abs(kg[k] - ko[k]).max() < 1e-8, k
    assert np.array_equal(kg["P"][:4], w["kf"]["P"][:4]) and np.array_equal(kg["q"][:4], w["kf"]["q"][:4])      # fixed keyframes stay put
    assert np.abs(g.get_points() - o.get_points()).max() < 1e-7
    cg, co = g.cull_observations(), o.cull_observations()
    assert np.array_equal(cg["bad_points"], co["bad_points"]) and np.array_equal(cg["bad_lines"], co["bad_lines"])
    g.close(); o.close()


@pytest.mark.gpu
def test_use_marg_off_call_site_through_the_facade(pkg, hip, tmp_path):
    """the same graph built by the call-site-shaped code of tools/localba_harness.cpp (`nomarg`): vertices in the reference's
    insertion order (window keyframes first, then the fixed ones), culling loop with computeError() on the gated edges"""
    w = make_nomarg_window(pkg)
    K = len(w["kf"]["P"])
    exe = build_harness()
    win, res = str(tmp_path / "w.bin"), str(tmp_path / "r.bin")
    write_window(w, win, do_marg=0)
    with open(win, "ab") as f:
        w["role"].astype(np.int32).tofile(f); w["imu"]["kf_i"].astype(np.int32).tofile(f); w["imu"]["kf_j"].astype(np.int32).tofile(f)
    subprocess.check_call([exe, "nomarg", win, res], timeout=120)
    Np, Nl, Ep, El = len(w["points"]), len(w["lines"]), len(w["po_pt"]), len(w["lo_ln"])
    r = read_result(res, K, Np, Nl)
    with open(res, "rb") as f:
        f.seek(-(Ep + El), 2)
        bad = np.fromfile(f, np.uint8, Ep + El)
    g = pkg.new_problem(); g.upload_window(w)
    out = pkg.protocol.local_ba(g)
    ref = pkg.protocol.results(g)
    assert r["gated"] == out["gated"]
    for k in ("P", "V", "q", "points", "lines"):
        assert np.abs(r[k] - ref[k]).max() < 1e-9, k
    win_kf = w["role"] == 0
    assert np.abs(r["dbg"][win_kf] - ref["dbg"][win_kf]).max() < 1e-9
    c = g.cull_observations()
    assert np.array_equal(bad[:Ep].astype(bool), c["bad_points"]) and np.array_equal(bad[Ep:].astype(bool), c["bad_lines"])
    g.close()
