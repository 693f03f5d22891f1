"""tests/golden/mp_vectors.json — the per-edge arithmetic of the path evaluated with mpmath at 40 digits from the reference's source
text (tests/golden/make_mp_vectors.py: an implementation independent of the oracle and of the kernels) — against
  (1) the CPU oracle's per-edge evaluators,
  (2) the device formulas (csrc/plba_math.h) compiled for the host,
  (3) the HIP kernels, through windows assembled from the same inputs (-m gpu).
This is the independent pin SURVEY 8(c) prescribes in place of reference-held fixtures: SO3 exp / log / Jr / Jr^-1 across the
reference's thresholds (theta < 1e-10, < 1e-5, near pi, negative real part), point / line / IMU PVR / bias / prior residuals and the
point / line Jacobians (incl. SURVEY B-Q1), the NavState oplus."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pl-inertial-slam_amd", "csrc")
SO = os.path.join(CSRC, "_obj", "libplba_math_hostcheck.so")
dp = C.POINTER(C.c_double)
TOL = 2e-13      # relative to the magnitude of the quantity (fp64 rounding of ~100-operation formulas)


@pytest.fixture(scope="module")
def vec():
    with open(os.path.join(ROOT, "tests", "golden", "mp_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def hc():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    src = os.path.join(CSRC, "plba_math_hostcheck.cpp")
    hdr = os.path.join(CSRC, "plba_math.h")
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", CSRC, "-o", SO, src])
    return C.CDLL(SO)


def _d(a):
    return a.ctypes.data_as(dp)


def _close(a, b, tol=TOL, what=""):
    a, b = np.asarray(a, float), np.asarray(b, float)
    sc = max(np.abs(b).max(), 1e-300)
    assert np.abs(a - b).max() <= tol * max(sc, 1.0) if sc < 1 else np.abs(a - b).max() <= tol * sc, (what, np.abs(a - b).max(), sc)


def _nav(n):
    return np.array(n["P"] + n["V"] + n["q"] + n["bg"] + n["ba"] + n["dbg"] + n["dba"], float)


def _camv(c):
    return np.array([c["fx"], c["fy"], c["cx"], c["cy"]] + list(np.ravel(c["Rbc"])) + list(c["Pbc"]), float)


def _pre142(pre):
    o = np.zeros(142)
    o[0:3] = pre["dP"]; o[3:6] = pre["dV"]; o[6:15] = np.ravel(pre["dR"]); o[15:24] = np.ravel(pre["JPg"]); o[24:33] = np.ravel(pre["JPa"])
    o[33:42] = np.ravel(pre["JVg"]); o[42:51] = np.ravel(pre["JVa"]); o[51:60] = np.ravel(pre["JRg"]); o[60:141] = np.eye(9).ravel() * 1e-4; o[141] = pre["dt"]
    return o


# ---- (1) the oracle -------------------------------------------------------------------------------------------------------------------
def test_oracle_so3(vec, orc):
    for v in vec["so3"]:
        if "w" in v:
            w = np.array(v["w"])
            _close(orc.so3_exp(w), v["exp"], what="exp")
            _close(orc.so3_log(np.array(v["exp"])), v["log_of_exp"], tol=1e-12, what="log(exp)")      # conditioned like 1 / sin near pi
            # (1 - cos theta) / theta and 1 - sin theta / theta cancel in fp64 just above the 1e-5 threshold (absolute ~1e-11: the reference's own
            # formula, shared by every fp64 implementation); (1 + cos) theta / (2 sin) is conditioned like 1 / sin near pi
            _close(orc.so3_jr(w), v["Jr"], tol=5e-11, what="Jr"); _close(orc.so3_jrinv(w), v["JrInv"], tol=5e-11, what="JrInv")
        else:
            _close(orc.so3_log(np.array(v["q"])), v["log"], tol=1e-12, what="log")


def test_oracle_point_and_line_edges(vec, orc):
    cam = _camv(vec["cam"])
    for v in vec["point"]:
        e, Ji, Jj, dpos = orc.eval_point_edge(cam, _nav(v["nav"]), np.array(v["Pw"]), np.array(v["obs"]))
        _close(e, v["e"], tol=1e-12, what="point e"); assert dpos == v["depth_positive"]
        sc = np.abs(v["JdR"]).max()
        assert np.abs(Ji - np.array(v["Jl"])).max() < 1e-12 * sc and np.abs(Jj[:, 0:3] - np.array(v["JdP"])).max() < 1e-12 * sc
        assert np.abs(Jj[:, 6:9] - np.array(v["JdR"])).max() < 1e-12 * sc and np.abs(Jj[:, 3:6]).max() == 0.0
    for v in vec["line"]:
        e, Ji, Jj, dpos = orc.eval_line_edge(cam, _nav(v["nav"]), np.array(v["L"]), np.array(v["obs"]))
        _close(e[:2], v["e"], tol=1e-12, what="line e"); assert e[2] == 0.0 and dpos == v["depth_positive"]
        sc = max(np.abs(v["JR_s"]).max(), np.abs(v["JR_e"]).max())
        assert np.abs(Ji[0, 0:3] - np.array(v["Jl_s"])[0]).max() < 1e-12 * sc and np.abs(Ji[1, 3:6] - np.array(v["Jl_e"])[0]).max() < 1e-12 * sc
        assert np.abs(Ji[0, 3:6]).max() == 0.0 and np.abs(Ji[1, 0:3]).max() == 0.0
        assert np.abs(Jj[0, 0:3] - np.array(v["JP_s"])[0]).max() < 1e-12 * sc and np.abs(Jj[1, 0:3] - np.array(v["JP_e"])[0]).max() < 1e-12 * sc      # B-Q1: the world-frame block
        assert np.abs(Jj[0, 6:9] - np.array(v["JR_s"])[0]).max() < 1e-12 * sc and np.abs(Jj[1, 6:9] - np.array(v["JR_e"])[0]).max() < 1e-12 * sc


def test_oracle_imu_edges_oplus_and_prior(vec, orc, hc):
    for v in vec["pvr"]:
        e, _, _, _ = orc.eval_pvr_edge(np.array(v["gw"]), _nav(v["navi"]), _nav(v["navj"]), _nav(v["navi"]), _pre142(v["pre"]))
        _close(e, v["e"], tol=1e-12, what="pvr e")
    for v in vec["oplus"]:
        o = orc.nav_oplus_pvr(_nav(v["nav"]), np.array(v["u"]))
        _close(o[:10], v["out"], tol=1e-13, what="oplus")


# ---- (2) the device formulas compiled for the host --------------------------------------------------------------------------------------
def test_device_formulas(vec, hc):
    cam = _camv(vec["cam"])
    for v in vec["so3"]:
        if "w" in v:
            w = np.array(v["w"]); q, J, lg = np.zeros(4), np.zeros(9), np.zeros(3)
            hc.hc_so3_exp(_d(w), _d(q)); _close(q, v["exp"], what="exp")
            e = np.array(v["exp"]); hc.hc_so3_log(_d(e), _d(lg)); _close(lg, v["log_of_exp"], tol=1e-12, what="log(exp)")
            hc.hc_so3_jr(_d(w), _d(J)); _close(J.reshape(3, 3), v["Jr"], tol=5e-11, what="Jr")
            hc.hc_so3_jrinv(_d(w), _d(J)); _close(J.reshape(3, 3), v["JrInv"], tol=5e-11, what="JrInv")
        else:
            q, lg = np.array(v["q"]), np.zeros(3)
            hc.hc_so3_log(_d(q), _d(lg)); _close(lg, v["log"], tol=1e-12, what="log")
    for v in vec["point"]:
        nav, Pw, obs = _nav(v["nav"]), np.array(v["Pw"]), np.array(v["obs"])
        e2, Jp, Jl, d = np.zeros(2), np.zeros(12), np.zeros(6), C.c_int()
        hc.hc_point_edge(_d(cam), _d(nav), _d(Pw), _d(obs), _d(e2), _d(Jp), _d(Jl), C.byref(d))
        sc = np.abs(v["JdR"]).max()
        _close(e2, v["e"], tol=1e-12, what="point e")
        assert np.abs(Jl.reshape(2, 3) - np.array(v["Jl"])).max() < 1e-12 * sc
        assert np.abs(Jp.reshape(2, 6)[:, :3] - np.array(v["JdP"])).max() < 1e-12 * sc and np.abs(Jp.reshape(2, 6)[:, 3:] - np.array(v["JdR"])).max() < 1e-12 * sc
    for v in vec["line"]:
        nav, L, l = _nav(v["nav"]), np.array(v["L"]), np.array(v["obs"])
        e2, Jp, Jl, d = np.zeros(2), np.zeros(12), np.zeros(6), C.c_int()
        hc.hc_line_edge(_d(cam), _d(nav), _d(L), _d(l), 0, _d(e2), _d(Jp), _d(Jl), C.byref(d))
        sc = max(np.abs(v["JR_s"]).max(), np.abs(v["JR_e"]).max())
        _close(e2, v["e"], tol=1e-12, what="line e")
        assert np.abs(Jl[0:3] - np.array(v["Jl_s"])[0]).max() < 1e-12 * sc and np.abs(Jl[3:6] - np.array(v["Jl_e"])[0]).max() < 1e-12 * sc
        Jp = Jp.reshape(2, 6)
        assert np.abs(Jp[0, :3] - np.array(v["JP_s"])[0]).max() < 1e-12 * sc and np.abs(Jp[1, :3] - np.array(v["JP_e"])[0]).max() < 1e-12 * sc
        assert np.abs(Jp[0, 3:] - np.array(v["JR_s"])[0]).max() < 1e-12 * sc and np.abs(Jp[1, 3:] - np.array(v["JR_e"])[0]).max() < 1e-12 * sc
    for v in vec["pvr"]:
        e9, j0, j1, j2 = np.zeros(9), np.zeros(81), np.zeros(81), np.zeros(54)
        gw, ni, nj, pre = np.array(v["gw"]), _nav(v["navi"]), _nav(v["navj"]), _pre142(v["pre"])
        hc.hc_pvr_edge(_d(gw), _d(ni), _d(nj), _d(pre), _d(e9), _d(j0), _d(j1), _d(j2))
        _close(e9, v["e"], tol=1e-12, what="pvr e")
    for v in vec["bias"]:
        e6 = np.zeros(6); ni, nj = _nav(v["navi"]), _nav(v["navj"])
        hc.hc_bias_error(_d(ni), _d(nj), _d(e6)); _close(e6, v["e"], tol=1e-13, what="bias e")
    for v in vec["oplus"]:
        o = np.zeros(22); n, u = _nav(v["nav"]), np.array(v["u"])
        hc.hc_oplus_pvr(_d(n), _d(u), _d(o)); _close(o[:10], v["out"], tol=1e-13, what="oplus")
    for v in vec["prior_dx"]:
        dx = np.zeros(9); n, x0 = _nav(v["nav"]), np.array(v["x0"])
        hc.hc_prior_dx_pvr(_d(n), _d(x0), _d(dx)); _close(dx, v["dx"], tol=1e-12, what="prior dx")


# ---- (3) the HIP kernels ----------------------------------------------------------------------------------------------------------------
def _window(pkg, vec, navs, fused):
    """a window of len(navs) keyframes holding the vector states verbatim"""
    c = vec["cam"]
    K = len(navs)
    kf = dict(vid_pvr=(2 * np.arange(K)).astype(np.int32), vid_bias=(2 * np.arange(K) + 1).astype(np.int32),
              P=np.array([n["P"] for n in navs]), V=np.array([n["V"] for n in navs]), q=np.array([n["q"] for n in navs]),
              bg=np.array([n["bg"] for n in navs]), ba=np.array([n["ba"] for n in navs]), dbg=np.array([n["dbg"] for n in navs]), dba=np.array([n["dba"] for n in navs]),
              fixed_pvr=np.zeros(K, np.uint8), fixed_bias=np.zeros(K, np.uint8))
    return dict(cam=dict(fx=c["fx"], fy=c["fy"], cx=c["cx"], cy=c["cy"], Rbc=np.array(c["Rbc"]), Pbc=np.array(c["Pbc"])), gw=np.array([0, 0, -9.81]), kf=kf,
                points=np.zeros((0, 3)), lines=np.zeros((0, 6)), po_pt=np.zeros(0, np.int32), po_kf=np.zeros(0, np.int32), po_uv=np.zeros((0, 2)), po_w=np.zeros(0),
                lo_ln=np.zeros(0, np.int32), lo_kf=np.zeros(0, np.int32), lo_l=np.zeros((0, 3)), lo_w=np.zeros(0), imu=None, prior=None, huber={0: 2.4476, 1: 2.4476})


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [0, 2])
def test_hip_point_and_line_residuals(pkg, hip, vec, fused):
    """every point / line vector observed from its own keyframe (twice: a landmark needs two edges to be well posed), residuals read
    back from the device: the record-based linearisation pass (lm_fused = 0) and the fused landmark-major pass (2)"""
    navs = [v["nav"] for v in vec["point"]]
    w = _window(pkg, vec, navs, fused)
    n = len(navs)
    w["points"] = np.array([v["Pw"] for v in vec["point"]])
    w["po_pt"] = np.repeat(np.arange(n), 2).astype(np.int32); w["po_kf"] = np.stack([np.arange(n), (np.arange(n) + 1) % n], 1).ravel().astype(np.int32)
    w["po_uv"] = np.repeat(np.array([v["obs"] for v in vec["point"]]), 2, axis=0); w["po_w"] = np.ones(2 * n)
    w["lines"] = np.array([v["L"] for v in vec["line"]])
    w["lo_ln"] = np.repeat(np.arange(n), 2).astype(np.int32); w["lo_kf"] = w["po_kf"].copy()
    w["lo_l"] = np.repeat(np.array([v["obs"] for v in vec["line"]]), 2, axis=0); w["lo_w"] = np.ones(2 * n)
    # the line vectors carry their own keyframe states: same generator order as the point vectors (one state per index)
    assert all(vl["nav"] == vp["nav"] for vl, vp in zip(vec["line"], vec["point"]))
    g = pkg.new_problem(lm_fused=fused, chain_elim=1); g.upload_window(w)
    g.debug_build(1.0, False)
    assert g.debug_get("lm_fused")[0] == (1 if fused else 0)
    ep = g.debug_get("err_pt").reshape(-1, 2)[0::2]; el = g.debug_get("err_ln").reshape(-1, 3)[0::2]
    for k in range(n):
        _close(ep[k], vec["point"][k]["e"], tol=1e-12, what="hip point e")
        _close(el[k][:2], vec["line"][k]["e"], tol=1e-12, what="hip line e")
    g.close()


@pytest.mark.gpu
def test_hip_imu_residuals(pkg, hip, vec):
    """every IMU vector as a two-keyframe window: err_pvr / err_bias of k_linearize's pose-edge blocks"""
    for v in vec["pvr"]:
        w = _window(pkg, vec, [v["navi"], v["navj"]], 0)
        pre = _pre142(v["pre"])
        w["imu"] = dict(kf_i=np.array([0], np.int32), kf_j=np.array([1], np.int32), preint=pre[None, :], info_pvr=np.eye(9).reshape(1, 81), info_bias=np.eye(6).reshape(1, 36))
        w["huber"] = {}
        g = pkg.new_problem(); g.upload_window(w)
        g.debug_build(1.0, False)
        _close(g.debug_get("err_pvr"), v["e"], tol=1e-12, what="hip pvr e")
        b = next(x for x in vec["bias"] if x["navi"] == v["navi"] and x["navj"] == v["navj"])
        _close(g.debug_get("err_bias"), b["e"], tol=1e-13, what="hip bias e")
        g.close()
