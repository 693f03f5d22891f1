"""API-surface types (SURVEY §8a footnote: named by the task, no local-BA call site): the host-evaluated classes of
include/plba_g2o/{se3quat,types_sba,types_six_dof_expmap}.h against the oracle's restatement of
IMU/se3quat.h / IMU/types_six_dof_expmap.{h,cpp} and against finite differences through the vertices' own oplus."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp = C.POINTER(C.c_double)


def _d(a):
    return a.ctypes.data_as(dp)


@pytest.fixture(scope="module")
def shim(pkg, hip_lib_path):
    out_dir = os.path.join(ROOT, "tools", "_build_api_shim")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libapi_shim.so")
    src = os.path.join(ROOT, "tools", "api_surface_shim.cpp")
    pkgdir = os.path.dirname(hip_lib_path)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wno-unknown-pragmas", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(pkgdir, "csrc"), src, "-o", so, "-L", pkgdir, "-lplba_hip", "-Wl,-rpath," + pkgdir]
    subprocess.run(cmd, check=True, capture_output=True)
    lib = C.CDLL(so)
    for f in ("shim_se3_exp", "shim_se3_log", "shim_se3_oplus", "shim_se3_inverse_mul", "shim_se3_vertex_io", "shim_eval_se3_edge", "shim_eval_edge_se3", "shim_lstsq3"):
        getattr(lib, f).restype = None
    return lib


def _rand_se3(rng, orc):
    u = np.concatenate([rng.normal(size=3) * 0.6, rng.normal(size=3) * 2.0])
    q, t = np.zeros(4), np.zeros(3)
    orc.lib().cdll.orc_se3_exp(_d(u), _d(q), _d(t))
    return q, t


def test_se3quat_exp_log_oplus(shim, orc):
    rng = np.random.default_rng(3)
    o = orc.lib().cdll
    for k in range(40):
        u = np.concatenate([rng.normal(size=3), rng.normal(size=3) * 3.0])
        if k % 7 == 0:
            u[:3] *= 1e-7                      # the theta < 1e-5 branch (R = I + W + W^2, IMU/se3quat.h:237-243)
        qs, ts, qo, to = np.zeros(4), np.zeros(3), np.zeros(4), np.zeros(3)
        shim.shim_se3_exp(_d(u), _d(qs), _d(ts)); o.orc_se3_exp(_d(u), _d(qo), _d(to))
        assert np.allclose(qs, qo, atol=1e-15) and np.allclose(ts, to, atol=1e-14)
        assert qs[3] >= 0 and abs(np.linalg.norm(qs) - 1) < 1e-15
        ls, lo = np.zeros(6), np.zeros(6)
        shim.shim_se3_log(_d(qs), _d(ts), _d(ls)); o.orc_se3_log(_d(qo), _d(to), _d(lo))
        assert np.allclose(ls, lo, atol=1e-13)
        if np.linalg.norm(u[:3]) < 3.0 and k % 7:
            assert np.allclose(ls, u, atol=1e-9)                        # log(exp(u)) = u away from pi
        q, t = _rand_se3(rng, orc)
        d = rng.normal(size=6) * 0.05
        a, b, c, e = np.zeros(4), np.zeros(3), np.zeros(4), np.zeros(3)
        shim.shim_se3_oplus(_d(q), _d(t), _d(d), _d(a), _d(b)); o.orc_se3_oplus(_d(q), _d(t), _d(d), _d(c), _d(e))
        assert np.allclose(a, c, atol=1e-15) and np.allclose(b, e, atol=1e-14)
        shim.shim_se3_inverse_mul(_d(q), _d(t), _d(a), _d(b))
        assert np.allclose(a, [0, 0, 0, 1], atol=1e-15) and np.allclose(b, 0, atol=1e-14)
        shim.shim_se3_vertex_io(_d(q), _d(t), _d(a), _d(b))            # write (camera -> world) then read back
        assert np.allclose(a, q, atol=1e-15) and np.allclose(b, t, atol=1e-13)


CAM = np.array([458.654, 457.296, 367.215, 248.375, 47.9])


def _eval(shim, kind, q, t, X, obs):
    D = 3 if kind in (1, 3) else 2
    err, Jp, Jx = np.zeros(D), np.zeros((D, 3)), np.zeros((D, 6))
    dpos, chi = C.c_int(0), C.c_double(0)
    shim.shim_eval_se3_edge(kind, _d(CAM), _d(q), _d(t), _d(X), _d(obs), _d(err), _d(Jp), _d(Jx), C.byref(dpos), C.byref(chi))
    return err, Jp, Jx, bool(dpos.value), chi.value


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_se3_projection_edges(shim, orc, kind):
    rng = np.random.default_rng(10 + kind)
    o = orc.lib().cdll
    D = 3 if kind in (1, 3) else 2
    for _ in range(25):
        q, t = _rand_se3(rng, orc)
        Pc = np.array([rng.uniform(-2, 2), rng.uniform(-1.5, 1.5), rng.uniform(1.0, 8.0)])
        R = orc.quat_to_R(q)
        X = R.T @ (Pc - t)
        obs = rng.normal(size=3) * 3 + np.array([CAM[0] * Pc[0] / Pc[2] + CAM[2], CAM[1] * Pc[1] / Pc[2] + CAM[3], CAM[0] * Pc[0] / Pc[2] + CAM[2] - CAM[4] / Pc[2]])
        err, Jp, Jx, dpos, chi = _eval(shim, kind, q, t, X, obs)
        eo, Jpo, Jxo, dpo = np.zeros(D), np.zeros((D, 3)), np.zeros((D, 6)), C.c_int(0)
        o.orc_eval_se3_edge(kind, _d(CAM), _d(q), _d(t), _d(X), _d(obs), _d(eo), _d(Jpo), _d(Jxo), C.byref(dpo))
        assert np.allclose(err, eo, rtol=0, atol=1e-10) and np.allclose(Jx, Jxo, rtol=1e-12, atol=1e-12) and dpos == bool(dpo.value) and dpos
        if kind < 2:
            assert np.allclose(Jp, Jpo, rtol=1e-12, atol=1e-12)
        assert chi == pytest.approx(2.0 * float(err @ err), rel=1e-14)
        # finite differences through the vertices' own updates (point: +=, pose: exp(delta) * T); the stereo rows see
        # the float rounding of 1/z, hence the step and tolerance
        h = 1e-6 if D == 2 else 2e-3
        tol = 1e-5 if D == 2 else 2e-2
        scale = max(1.0, np.abs(Jx).max())
        for c in range(6):
            d = np.zeros(6); d[c] = h
            qa, ta, qb, tb = np.zeros(4), np.zeros(3), np.zeros(4), np.zeros(3)
            shim.shim_se3_oplus(_d(q), _d(t), _d(d), _d(qa), _d(ta)); shim.shim_se3_oplus(_d(q), _d(t), _d(-d), _d(qb), _d(tb))
            fd = (_eval(shim, kind, qa, ta, X, obs)[0] - _eval(shim, kind, qb, tb, X, obs)[0]) / (2 * h)
            assert np.abs(fd - Jx[:, c]).max() < tol * scale, (kind, c)
        if kind < 2:
            for c in range(3):
                d = np.zeros(3); d[c] = h
                fd = (_eval(shim, kind, q, t, X + d, obs)[0] - _eval(shim, kind, q, t, X - d, obs)[0]) / (2 * h)
                assert np.abs(fd - Jp[:, c]).max() < tol * scale, (kind, c)
    # behind the camera
    q, t = np.array([0, 0, 0, 1.0]), np.zeros(3)
    assert _eval(shim, kind, q, t, np.array([0.1, 0.1, -2.0]), np.zeros(3))[3] is False


def test_stereo_projection_keeps_the_float_inverse_depth(shim, orc):
    """IMU/types_six_dof_expmap.cpp:150-157, 299-306: invz is a float; for the two-vertex edge so is bf * invz"""
    q, t = np.array([0, 0, 0, 1.0]), np.zeros(3)
    X = np.array([0.3, -0.2, 3.0])
    obs = np.zeros(3)
    invz = np.float32(1.0 / X[2])
    for kind in (1, 3):
        err = _eval(shim, kind, q, t, X, obs)[0]
        u = X[0] * float(invz) * CAM[0] + CAM[2]
        ur = u - (float(np.float32(CAM[4]) * invz) if kind == 1 else CAM[4] * float(invz))
        assert err[0] == -u and err[2] == -ur
        assert err[0] != -(X[0] / X[2] * CAM[0] + CAM[2])       # a double 1/z would give a different pixel


# ---- API-surface types of IMU/g2otypes.h ---------------------------------------------------------------------------
def _cam_vec(pkg, orc):
    w = pkg.window
    return orc.cam_vec(dict(fx=w.FX, fy=w.FY, cx=w.CX, cy=w.CY, Rbc=w.T_BS[:3, :3], Pbc=w.T_BS[:3, 3]))


def _nav_in_front(pkg, orc, rng):
    """a keyframe state and a world point 1-8 m in front of its camera"""
    w = pkg.window
    Rwb = w.exp_so3(rng.normal(size=3) * 0.8)
    Pwb = rng.normal(size=3)
    nav = orc.nav_vec(Pwb, rng.normal(size=3), w.quat_from_R(Rwb))
    Rbc, Pbc = w.T_BS[:3, :3], w.T_BS[:3, 3]
    Pc = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1, 1), rng.uniform(1, 8)])
    Pw = Rwb @ (Rbc @ Pc + Pbc) + Pwb
    return nav, Pw


def test_pvr_point_only_pose_edge(shim, orc, pkg):
    """EdgeNavStatePVRPointXYZOnlyPose == the pose block of EdgeNavStatePVRPointXYZ (IMU/g2otypes.cpp:343-394 vs 286-341)"""
    rng = np.random.default_rng(21)
    cam = _cam_vec(pkg, orc)
    shim.shim_eval_pvr_point_onlypose.restype = None
    for _ in range(20):
        nav, Pw = _nav_in_front(pkg, orc, rng)
        obs = rng.uniform(100, 400, size=2)
        e, J, dpos = np.zeros(2), np.zeros((2, 9)), C.c_int(0)
        shim.shim_eval_pvr_point_onlypose(_d(cam), _d(nav), _d(Pw), _d(obs), _d(e), _d(J), C.byref(dpos))
        eo, Ji, Jj, dpo = orc.eval_point_edge(cam, nav, Pw, obs)
        assert np.allclose(e, eo, atol=1e-10) and np.allclose(J, Jj.reshape(2, 9), rtol=1e-11, atol=1e-11) and dpos.value == 1 == int(dpo)
        assert np.all(J[:, 3:6] == 0)


def test_line_point_edge(shim, orc, pkg):
    """EdgeNavStateLinePoint (IMU/g2otypes.h:919-1000, .cpp:1383-1421) == row 0 of the two-end-point line edge; the
    reference's own test/test.cpp scenario is the known answer (603, 0, 0)"""
    rng = np.random.default_rng(22)
    cam = _cam_vec(pkg, orc)
    shim.shim_eval_linepoint.restype = None
    for _ in range(20):
        nav, Pw = _nav_in_front(pkg, orc, rng)
        l = rng.normal(size=3); l /= np.linalg.norm(l[:2])
        e, Ji, Jj, dpos = np.zeros(3), np.zeros((3, 3)), np.zeros((3, 9)), C.c_int(0)
        shim.shim_eval_linepoint(_d(cam), _d(nav), _d(Pw), _d(l), _d(e), _d(Ji), _d(Jj), C.byref(dpos))
        assert np.allclose(e, orc.eval_linepoint_edge(cam, nav, Pw, l), atol=1e-10) and e[1] == 0 and e[2] == 0
        eo, Jio, Jjo, _ = orc.eval_line_edge(cam, nav, np.concatenate([Pw, Pw]), l, fix_q1=0)
        assert np.allclose(Ji[0], Jio.reshape(3, 6)[0, :3], rtol=1e-11, atol=1e-11)
        assert np.allclose(Jj[0], Jjo.reshape(3, 9)[0], rtol=1e-11, atol=1e-11)
        assert np.all(Ji[1:] == 0) and np.all(Jj[1:] == 0) and dpos.value == 1
    # /root/reference/test/test.cpp:15-48: identity pose and extrinsics, fx=fy=cx=cy=100, P=(5,5,5), measurement (1,2,3)
    camt = orc.cam_vec(dict(fx=100.0, fy=100.0, cx=100.0, cy=100.0, Rbc=np.eye(3), Pbc=np.zeros(3)))
    navt = orc.nav_vec(np.zeros(3), np.zeros(3), np.array([0, 0, 0, 1.0]))
    e, Ji, Jj, dpos = np.zeros(3), np.zeros((3, 3)), np.zeros((3, 9)), C.c_int(0)
    shim.shim_eval_linepoint(_d(camt), _d(navt), _d(np.array([5.0, 5.0, 5.0])), _d(np.array([1.0, 2.0, 3.0])), _d(e), _d(Ji), _d(Jj), C.byref(dpos))
    assert np.array_equal(e, [603.0, 0.0, 0.0])


def test_gyr_bias_edge(shim, orc, pkg):
    rng = np.random.default_rng(23)
    w = pkg.window
    o = orc.lib().cdll
    shim.shim_eval_gyrbias.restype = None
    for _ in range(20):
        Ri, Rj = w.exp_so3(rng.normal(size=3)), None
        dR = w.exp_so3(rng.normal(size=3) * 0.3)
        Rj = Ri @ dR @ w.exp_so3(rng.normal(size=3) * 0.02)
        Jg = -np.eye(3) * 0.25 + rng.normal(size=(3, 3)) * 0.01
        bg = rng.normal(size=3) * 0.01
        args = [np.ascontiguousarray(a) for a in (dR, Jg, Ri, Rj, bg)]
        e, J, eo, Jo = np.zeros(3), np.zeros((3, 3)), np.zeros(3), np.zeros((3, 3))
        shim.shim_eval_gyrbias(*[_d(a) for a in args], _d(e), _d(J))
        o.orc_eval_gyrbias_edge(*[_d(a) for a in args], _d(eo), _d(Jo))
        assert np.allclose(e, eo, atol=1e-13) and np.allclose(J, Jo, rtol=1e-12, atol=1e-13)
        # the analytic Jacobian is the derivative at bg = 0 (IMU/g2otypes.cpp:1277-1287 drops the bias term)
        z = np.zeros(3)
        fd = np.zeros((3, 3))
        for c in range(3):
            h = np.zeros(3); h[c] = 1e-6
            ep, em = np.zeros(3), np.zeros(3)
            shim.shim_eval_gyrbias(*[_d(a) for a in args[:4]], _d(z + h), _d(ep), _d(J))
            shim.shim_eval_gyrbias(*[_d(a) for a in args[:4]], _d(z - h), _d(em), _d(J))
            fd[:, c] = (ep - em) / 2e-6
        assert np.allclose(fd, J, atol=1e-6)


# ---- the 15-DoF family (IMU/g2otypes.h:411-695) ------------------------------------------------------------------------
def _imu_pair(pkg, orc, rng):
    """two consecutive full states roughly consistent with a 0.25 s preintegrated measurement"""
    w = pkg.window
    S = 50
    om = rng.normal(size=(1, S, 3)) * 0.2
    ac = rng.normal(size=(1, S, 3)) * 1.0 + np.array([0, 0, 9.81])
    pre = w.preintegrate(om, ac, 0.005)[0]
    Ri = w.exp_so3(rng.normal(size=3) * 0.5)
    Pi, Vi = rng.normal(size=3), rng.normal(size=3)
    gw = np.array([0, 0, -9.81])
    dT = pre[141]
    dR = pre[6:15].reshape(3, 3)
    Pj = Pi + Vi * dT + 0.5 * gw * dT * dT + Ri @ pre[0:3] + rng.normal(size=3) * 0.02
    Vj = Vi + gw * dT + Ri @ pre[3:6] + rng.normal(size=3) * 0.02
    Rj = Ri @ dR @ w.exp_so3(rng.normal(size=3) * 0.02)
    bg, ba = rng.normal(size=3) * 1e-3, rng.normal(size=3) * 1e-2
    navi = orc.nav_vec(Pi, Vi, w.quat_from_R(Ri), bg, ba, rng.normal(size=3) * 1e-3, rng.normal(size=3) * 1e-2)
    navj = orc.nav_vec(Pj, Vj, w.quat_from_R(Rj), bg, ba, rng.normal(size=3) * 1e-3, rng.normal(size=3) * 1e-2)
    return gw, navi, navj, pre


def _nav_edge(shim, gw, navi, navj, pre, with_gw):
    e, Ji, Jj, Jg = np.zeros(15), np.zeros((15, 15)), np.zeros((15, 15)), np.zeros((15, 2))
    shim.shim_navstate_edge(_d(gw), _d(navi), _d(navj), _d(pre), int(with_gw), _d(e), _d(Ji), _d(Jj), _d(Jg))
    return e, Ji, Jj, Jg


def _oplus(shim, fn, nav, u):
    out = np.zeros(22)
    getattr(shim, fn)(_d(nav), _d(np.ascontiguousarray(u)), _d(out))
    return out


@pytest.mark.parametrize("with_gw", [False, True])
def test_navstate_imu_edges(shim, orc, pkg, with_gw):
    """EdgeNavState / EdgeNavStateGw: residual = (EdgeNavStatePVR, EdgeNavStateBias) of the on-path oracle; Jacobians
    against finite differences through VertexNavState::oplusImpl (and VertexGravityW's 2-DoF update)"""
    rng = np.random.default_rng(31 + with_gw)
    for f in ("shim_navstate_edge", "shim_navstate_oplus", "shim_gravity_oplus"):
        getattr(shim, f).restype = None
    for _ in range(6):
        gw, navi, navj, pre = _imu_pair(pkg, orc, rng)
        e, Ji, Jj, Jg = _nav_edge(shim, gw, navi, navj, pre, with_gw)
        e9 = orc.eval_pvr_edge(gw, navi, navj, navi, pre)[0]
        assert np.allclose(e[:9], e9, atol=1e-12)
        assert np.allclose(e[9:12], (navj[10:13] + navj[16:19]) - (navi[10:13] + navi[16:19]), atol=1e-15)
        assert np.allclose(e[12:15], (navj[13:16] + navj[19:22]) - (navi[13:16] + navi[19:22]), atol=1e-15)
        h = 1e-6
        for which, J in ((0, Ji), (1, Jj)):
            for c in range(15):
                u = np.zeros(15); u[c] = h
                a = [navi, navj]; b = [navi, navj]
                a[which] = _oplus(shim, "shim_navstate_oplus", a[which], u); b[which] = _oplus(shim, "shim_navstate_oplus", b[which], -u)
                fd = (_nav_edge(shim, gw, a[0], a[1], pre, with_gw)[0] - _nav_edge(shim, gw, b[0], b[1], pre, with_gw)[0]) / (2 * h)
                # the rotation-residual rows use the first-order JrInv forms of the paper: exact only for small residuals
                assert np.abs(fd - J[:, c]).max() < 2e-3 * max(1.0, np.abs(J[:, c]).max()), (which, c)
        if with_gw:
            for c in range(2):
                u = np.zeros(2); u[c] = h
                ga, gb = np.zeros(3), np.zeros(3)
                shim.shim_gravity_oplus(_d(gw), _d(u), _d(ga)); shim.shim_gravity_oplus(_d(gw), _d(-u), _d(gb))
                fd = (_nav_edge(shim, ga, navi, navj, pre, True)[0] - _nav_edge(shim, gb, navi, navj, pre, True)[0]) / (2 * h)
                assert np.abs(fd - Jg[:, c]).max() < 1e-6
            assert np.all(Jg[6:] == 0)
    # gravity vertex: 2-DoF rotation keeps the norm; origin is (0, 0, 9.81)
    g = np.zeros(3)
    shim.shim_gravity_oplus(_d(np.array([0, 0, 9.81])), _d(np.array([0.1, -0.2])), _d(g))
    assert abs(np.linalg.norm(g) - 9.81) < 1e-12 and abs(g[0]) > 0.1


def test_navstate_prior_edges(shim, orc, pkg):
    rng = np.random.default_rng(33)
    for f in ("shim_prior_edge", "shim_prior_pvrbias_edge", "shim_pvr_oplus", "shim_bias_oplus"):
        getattr(shim, f).restype = None
    w = pkg.window
    for _ in range(6):
        _, prior, _, _ = _imu_pair(pkg, orc, rng)
        est = _oplus(shim, "shim_navstate_oplus", prior, rng.normal(size=15) * 0.05)
        e, J = np.zeros(15), np.zeros((15, 15))
        shim.shim_prior_edge(_d(prior), _d(est), _d(e), _d(J))
        assert np.allclose(e[:6], prior[:6] - est[:6], atol=1e-15)
        Rp, Re = orc.quat_to_R(prior[6:10]), orc.quat_to_R(est[6:10])
        assert np.allclose(e[6:9], w.log_so3(Rp.T @ Re), atol=1e-10)
        assert np.allclose(e[9:15], (prior[10:16] + prior[16:22]) - (est[10:16] + est[16:22]), atol=1e-15)
        h = 1e-6
        for c in range(15):
            u = np.zeros(15); u[c] = h
            ea, eb, Jt = np.zeros(15), np.zeros(15), np.zeros((15, 15))
            shim.shim_prior_edge(_d(prior), _d(_oplus(shim, "shim_navstate_oplus", est, u)), _d(ea), _d(Jt))
            shim.shim_prior_edge(_d(prior), _d(_oplus(shim, "shim_navstate_oplus", est, -u)), _d(eb), _d(Jt))
            assert np.abs((ea - eb) / (2 * h) - J[:, c]).max() < 1e-6, c
        # the two-vertex form: same residual, Jacobian split 9 | 6
        e2, Ja, Jb = np.zeros(15), np.zeros((15, 9)), np.zeros((15, 6))
        shim.shim_prior_pvrbias_edge(_d(prior), _d(est), _d(est), _d(e2), _d(Ja), _d(Jb))
        assert np.array_equal(e2, e) and np.array_equal(Ja, J[:, :9]) and np.array_equal(Jb, J[:, 9:])


def test_navstate_point_edges(shim, orc, pkg):
    rng = np.random.default_rng(34)
    cam = _cam_vec(pkg, orc)
    shim.shim_navstate_point.restype = None
    for _ in range(10):
        nav, Pw = _nav_in_front(pkg, orc, rng)
        obs = rng.uniform(100, 400, size=2)
        eo, Jio, Jjo, _ = orc.eval_point_edge(cam, nav, Pw, obs)
        for only_pose in (0, 1):
            e, Ji, Jj, dpos = np.zeros(2), np.zeros((2, 3)), np.zeros((2, 15)), C.c_int(0)
            shim.shim_navstate_point(_d(cam), _d(nav), _d(Pw), _d(obs), only_pose, _d(e), _d(Ji), _d(Jj), C.byref(dpos))
            assert np.allclose(e, eo, atol=1e-10) and dpos.value == 1
            assert np.allclose(Jj[:, :9], Jjo, rtol=1e-11, atol=1e-11) and np.all(Jj[:, 9:] == 0)
            if not only_pose:
                assert np.allclose(Ji, Jio, rtol=1e-11, atol=1e-11)


def test_edge_se3_jacobians_are_the_derivatives_of_its_error(pkg, orc, shim):
    """include/plba_g2o/types_slam3d.h: analytic _jacobianOplusXi / Xj of EdgeSE3 against central differences of the oracle's
    error through the oracle's vertex update (tools/api_surface_shim.cpp evaluates the facade's edge)"""
    rng = np.random.default_rng(3)
    W = pkg.window
    lib = orc.lib().cdll
    for _ in range(5):
        X = []
        for _k in range(3):
            R = W.exp_so3(rng.normal(size=3) * 0.8); t = rng.normal(size=3) * 2
            X.append(np.ascontiguousarray(np.concatenate([R.ravel(), t])))
        Xi, Xj, Z = X
        e, Ji, Jj = np.zeros(6), np.zeros((6, 6)), np.zeros((6, 6))
        shim.shim_eval_edge_se3(_d(Xi), _d(Xj), _d(Z), _d(e), _d(Ji), _d(Jj))
        e0 = np.zeros(6)
        lib.orc_se3_edge_error(_d(Xi), _d(Xj), _d(Z), _d(e0))
        assert np.abs(e - e0).max() < 1e-12
        h = 1e-6
        for which, Jan in ((0, Ji), (1, Jj)):
            Jn = np.zeros((6, 6))
            for c in range(6):
                ep, em, Xp, Xm = np.zeros(6), np.zeros(6), np.zeros(12), np.zeros(12)
                u = np.zeros(6); u[c] = h
                lib.orc_se3_vertex_oplus(_d(Xj if which else Xi), _d(u), _d(Xp))
                u[c] = -h
                lib.orc_se3_vertex_oplus(_d(Xj if which else Xi), _d(u), _d(Xm))
                lib.orc_se3_edge_error(_d(Xi if which else Xp), _d(Xp if which else Xj), _d(Z), _d(ep))
                lib.orc_se3_edge_error(_d(Xi if which else Xm), _d(Xm if which else Xj), _d(Z), _d(em))
                Jn[:, c] = (ep - em) / (2 * h)
            assert np.abs(Jan - Jn).max() < 1e-7, which


def test_lstsq3_keeps_small_singular_values_and_truncates_noise(shim):
    """plba_vio::lstsq3 = JacobiSVD(A).solve(b) of tryVioInit's gravity / accelerometer-bias systems (src/mapHandler.cpp:4896, 4940).
    ADVICE r03: through A^T A a singular value below sqrt(eps) * sigma_max is rounding noise — with cond(A) = 1e9 it was inverted as
    if it were data.  The one-sided Jacobi on A resolves it (relative accuracy) and truncates only at Eigen's threshold."""
    rng = np.random.default_rng(11)
    rows = 30
    U, _ = np.linalg.qr(rng.normal(size=(rows, 3)))
    W, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    for sig, what in (([1.0, 1e-3, 1e-9], "cond 1e9, full rank"), ([1.0, 0.5, 1e-18], "numerically rank 2: the third direction must be dropped"),
                      ([3.0, 2.0, 1.0], "well conditioned")):
        A = (U * np.array(sig)) @ W.T
        xt = rng.normal(size=3)
        b = A @ xt + 1e-12 * rng.normal(size=rows)
        x = np.zeros(3)
        shim.shim_lstsq3(rows, _d(np.ascontiguousarray(A)), _d(b), _d(x))
        ref = np.linalg.lstsq(A, b, rcond=rows * np.finfo(float).eps)[0]      # SVD, the same truncation rule
        assert np.abs(x - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), what
        if sig[2] == 1e-18:
            assert np.abs(x).max() < 10.0      # a direction at rounding level inverted as data would give ~1e6
