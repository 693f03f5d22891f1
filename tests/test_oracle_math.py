"""CPU tests pinning the oracle's per-edge math by independent means:
known answers, finite differences against the vertex oplus, numpy cross-checks."""
import numpy as np
import pytest


def _cam(pkg):
    w = pkg.window
    return dict(fx=w.FX, fy=w.FY, cx=w.CX, cy=w.CY, Rbc=w.T_BS[:3, :3], Pbc=w.T_BS[:3, 3])


def _rand_nav(orc, pkg, rng, small_rot=False):
    R = pkg.window.exp_so3(rng.normal(size=3) * (1e-3 if small_rot else 0.7))
    q = pkg.window.quat_from_R(R)
    return orc.nav_vec(rng.normal(size=3), rng.normal(size=3), q, rng.normal(size=3) * 1e-2, rng.normal(size=3) * 1e-1,
                       rng.normal(size=3) * 1e-3, rng.normal(size=3) * 1e-2)


def test_known_answer_from_reference_test_cpp(orc):
    """/root/reference/test/test.cpp:15-48 with IMU/g2otypes.h:929-939: NavState identity, Rbc=I, Pbc=0,
    point (5,5,5), fx=fy=cx=cy=100, measurement (1,2,3) -> Pc=(5,5,5), uv=(200,200), error=(603,0,0)."""
    cam = np.array([100, 100, 100, 100, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=float)
    nav = orc.nav_vec(np.zeros(3), np.zeros(3), np.array([0, 0, 0, 1.0]))
    e = orc.eval_linepoint_edge(cam, nav, np.array([5.0, 5, 5]), np.array([1.0, 2, 3]))
    assert np.array_equal(e, [603.0, 0.0, 0.0])
    # the same geometry through the point edge: obs - proj
    e2, _, _, dpos = orc.eval_point_edge(cam, nav, np.array([5.0, 5, 5]), np.array([210.0, 190.0]))
    assert np.allclose(e2, [10.0, -10.0], atol=1e-12) and dpos


def test_so3_exp_log_roundtrip_and_small_angle(orc, pkg):
    rng = np.random.default_rng(0)
    for scale in (1.0, 1e-3, 1e-7, 1e-11, 0.0):
        w = rng.normal(size=3) * scale
        q = orc.so3_exp(w)
        assert abs(np.linalg.norm(q) - 1) < 1e-15
        assert np.allclose(orc.so3_log(q), w, atol=1e-12)
        assert np.allclose(orc.quat_to_R(q), pkg.window.exp_so3(w), atol=1e-12)
    # theta < 1e-10 Taylor branch constants (IMU/so3.cpp:265-270)
    w = np.array([3e-11, 0, 0])
    assert orc.so3_exp(w)[0] == pytest.approx(0.5 * 3e-11, rel=1e-12)


def test_so3_log_uses_atan_not_atan2(orc):
    """IMU/so3.cpp:243: 2*atan(n/w)/n — for w<0 the angle folds to (-pi/2,0), reproduced (B-Q13)."""
    th = 2.5
    q = np.array([np.sin(th / 2), 0, 0, np.cos(th / 2)])
    assert np.allclose(orc.so3_log(q), [th, 0, 0])
    qn = -q  # same rotation, w<0
    n, w = abs(qn[0]), qn[3]
    assert np.allclose(orc.so3_log(qn), 2 * np.arctan(n / w) / n * qn[:3])


def test_jr_jrinv(orc):
    rng = np.random.default_rng(1)
    for s in (0.5, 1e-2, 2e-5):
        w = rng.normal(size=3) * s
        assert np.allclose(orc.so3_jr(w) @ orc.so3_jrinv(w), np.eye(3), atol=1e-9)
    assert np.array_equal(orc.so3_jr(np.array([1e-6, 0, 0])), np.eye(3))      # theta < 1e-5 -> I
    assert np.array_equal(orc.so3_jrinv(np.array([0, 9e-6, 0])), np.eye(3))


def test_quat_matrix_conversions_follow_eigen(orc, pkg):
    rng = np.random.default_rng(2)
    for _ in range(50):
        R = pkg.window.exp_so3(rng.normal(size=3) * 2.5)
        q = orc.R_to_quat(R)
        assert np.allclose(orc.quat_to_R(q), R, atol=1e-12)
        assert np.allclose(q / np.linalg.norm(q), pkg.window.quat_from_R(R), atol=1e-12)
    # trace <= 0 branch
    R = pkg.window.exp_so3(np.array([np.pi - 1e-3, 0, 0]))
    assert np.allclose(orc.quat_to_R(orc.R_to_quat(R)), R, atol=1e-12)


def test_huber(orc):
    d = float(np.float32(np.sqrt(5.991)))
    assert np.array_equal(orc.huber(3.0, d), [3.0, 1.0, 0.0])
    r = orc.huber(50.0, d)
    assert r[0] == pytest.approx(2 * np.sqrt(50.0) * d - d * d) and r[1] == pytest.approx(d / np.sqrt(50.0))
    assert r[2] == pytest.approx(-0.5 * r[1] / 50.0)


def _fd(f, x0, n, oplus, h=1e-6):
    """central differences of f(oplus(x0, delta)) w.r.t. delta in R^n"""
    cols = []
    for i in range(n):
        d = np.zeros(n); d[i] = h
        cols.append((f(oplus(x0, d)) - f(oplus(x0, -d))) / (2 * h))
    return np.stack(cols, 1)


def test_point_edge_jacobians_fd(orc, pkg):
    rng = np.random.default_rng(3)
    cam = orc.cam_vec(_cam(pkg))
    for _ in range(5):
        nav = _rand_nav(orc, pkg, rng)
        R = orc.quat_to_R(nav[6:10])
        Pc = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(1.5, 6)])
        Pw = R @ (pkg.window.T_BS[:3, :3] @ Pc + pkg.window.T_BS[:3, 3]) + nav[0:3]
        obs = rng.uniform(0, 400, size=2)
        e, Ji, Jj, dpos = orc.eval_point_edge(cam, nav, Pw, obs)
        assert dpos
        fd_i = _fd(lambda P: orc.eval_point_edge(cam, nav, P, obs, jac=False)[0], Pw, 3, lambda x, d: x + d)
        fd_j = _fd(lambda s: orc.eval_point_edge(cam, s, Pw, obs, jac=False)[0], nav, 9, orc.nav_oplus_pvr)
        assert np.allclose(Ji, fd_i, rtol=1e-5, atol=1e-4)
        assert np.allclose(Jj, fd_j, rtol=1e-5, atol=1e-4)
        assert np.all(Jj[:, 3:6] == 0)


def test_line_edge_jacobians_fd_and_q1_defect(orc, pkg):
    """Landmark and rotation blocks are true derivatives; the position block is the reference's
    world-frame formula (IMU/g2otypes.cpp:1347-1348, SURVEY B-Q1) and matches finite differences only
    with fix_line_position_jacobian=1."""
    rng = np.random.default_rng(4)
    cam = orc.cam_vec(_cam(pkg))
    Rbc, Pbc = pkg.window.T_BS[:3, :3], pkg.window.T_BS[:3, 3]
    for _ in range(5):
        nav = _rand_nav(orc, pkg, rng)
        R = orc.quat_to_R(nav[6:10])
        Pcs = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(1.5, 6)])
        Pce = Pcs + rng.normal(size=3) * 0.3
        L = np.concatenate([R @ (Rbc @ Pcs + Pbc) + nav[0:3], R @ (Rbc @ Pce + Pbc) + nav[0:3]])
        l = rng.normal(size=3); l /= np.hypot(l[0], l[1])
        e, Ji, Jj, dpos = orc.eval_line_edge(cam, nav, L, l)
        assert e[2] == 0 and np.all(Ji[2] == 0) and np.all(Jj[2] == 0)
        fd_i = _fd(lambda x: orc.eval_line_edge(cam, nav, x, l, jac=False)[0], L, 6, lambda x, d: x + d)
        fd_j = _fd(lambda s: orc.eval_line_edge(cam, s, L, l, jac=False)[0], nav, 9, orc.nav_oplus_pvr)
        assert np.allclose(Ji, fd_i, rtol=1e-5, atol=1e-4)
        assert np.allclose(Jj[:, 6:9], fd_j[:, 6:9], rtol=1e-5, atol=1e-4)
        assert np.all(Jj[:, 3:6] == 0)
        # reference position block == world-frame derivative == FD(body-frame) @ R^T
        assert np.allclose(Jj[:, 0:3], fd_j[:, 0:3] @ R.T, rtol=1e-5, atol=1e-4)
        assert not np.allclose(Jj[:, 0:3], fd_j[:, 0:3], rtol=1e-3, atol=1e-3)
        _, _, Jfix, _ = orc.eval_line_edge(cam, nav, L, l, fix_q1=1)
        assert np.allclose(Jfix[:, 0:3], fd_j[:, 0:3], rtol=1e-5, atol=1e-4)


def _rand_preint(pkg, rng):
    S = 50
    w = rng.normal(size=(1, S, 3)) * 0.3
    a = rng.normal(size=(1, S, 3)) * 2 + np.array([0, 0, 9.8])
    return pkg.window.preintegrate(w, a, 0.005)[0]


def test_pvr_edge_jacobians_fd(orc, pkg):
    rng = np.random.default_rng(5)
    gw = np.array([0, 0, -9.81])
    for _ in range(4):
        pre = _rand_preint(pkg, rng)
        navi = _rand_nav(orc, pkg, rng)
        navj = navi.copy()
        # make j roughly consistent with the preintegration so the rotation residual is small but non-zero
        Ri = orc.quat_to_R(navi[6:10])
        dR = pre[6:15].reshape(3, 3)
        navj[6:10] = pkg.window.quat_from_R(Ri @ dR @ pkg.window.exp_so3(rng.normal(size=3) * 0.02))
        navj[0:3] = navi[0:3] + navi[3:6] * 0.25 + 0.5 * gw * 0.0625 + Ri @ pre[0:3] + rng.normal(size=3) * 0.01
        navj[3:6] = navi[3:6] + gw * 0.25 + Ri @ pre[3:6] + rng.normal(size=3) * 0.01
        navi[16:22] = rng.normal(size=6) * 1e-4
        e, J0, J1, J2 = orc.eval_pvr_edge(gw, navi, navj, navi, pre)
        f = lambda a, b, c: orc.eval_pvr_edge(gw, a, b, c, pre, jac=False)[0]
        # PVR_i and Bias_i live in two vertices holding copies of the NavState: perturb them separately
        fd0 = _fd(lambda s: f(s, navj, navi), navi, 9, orc.nav_oplus_pvr)
        fd1 = _fd(lambda s: f(navi, s, navi), navj, 9, orc.nav_oplus_pvr)
        fd2 = _fd(lambda s: f(navi, navj, s), navi, 6, orc.nav_oplus_bias)
        assert np.allclose(J0, fd0, rtol=1e-4, atol=2e-4)
        assert np.allclose(J1, fd1, rtol=1e-4, atol=2e-4)
        assert np.allclose(J2, fd2, rtol=1e-3, atol=2e-3)   # rotation/bias block is first order in dbg (Forster)


def test_oplus_conventions(orc, pkg):
    """IMU/NavState.cpp:69-121: P += R*dp (body frame), V += dv, R = R*Exp(dphi); bias update touches only the deltas."""
    rng = np.random.default_rng(6)
    nav = _rand_nav(orc, pkg, rng)
    u = rng.normal(size=9) * 0.1
    o = orc.nav_oplus_pvr(nav, u)
    R = orc.quat_to_R(nav[6:10])
    assert np.allclose(o[0:3], nav[0:3] + R @ u[0:3], atol=1e-14)
    assert np.allclose(o[3:6], nav[3:6] + u[3:6])
    assert np.allclose(orc.quat_to_R(o[6:10]), R @ pkg.window.exp_so3(u[6:9]), atol=1e-12)
    assert np.array_equal(o[10:], nav[10:])
    b = orc.nav_oplus_bias(nav, u[:6])
    assert np.array_equal(b[:16], nav[:16]) and np.allclose(b[16:19], nav[16:19] + u[0:3]) and np.allclose(b[19:22], nav[19:22] + u[3:6])


def test_sym_eig_matches_numpy(orc):
    rng = np.random.default_rng(7)
    for n in (1, 3, 17, 60):
        A = rng.normal(size=(n, n)); A = A @ A.T
        if n > 3:
            A[:, :2] = 0; A[:2, :] = 0   # rank deficient like a marginalization block
        w, V = orc.sym_eig(A)
        assert np.allclose(w, np.linalg.eigvalsh(A), atol=1e-10 * max(1, abs(w).max()))
        assert np.allclose(V @ np.diag(w) @ V.T, A, atol=1e-10 * max(1, abs(w).max()))
        assert np.allclose(V.T @ V, np.eye(n), atol=1e-12)


def test_preintegration_generator_matches_oracle_update(orc, pkg):
    """window.preintegrate (numpy) vs the C restatement of IMUPreintegrator::update (IMU/IMUPreintegrator.cpp:80-139)."""
    rng = np.random.default_rng(8)
    S = 20
    w = rng.normal(size=(1, S, 3)) * 0.5
    a = rng.normal(size=(1, S, 3)) * 3
    ref = pkg.window.preintegrate(w, a, 0.005)[0]
    pre = np.zeros(142); pre[6:15] = np.eye(3).ravel()
    for s in range(S):
        pre = orc.preint_update(pre, w[0, s], a[0, s], 0.005, pkg.window.GYR_MEAS_COV, pkg.window.ACC_MEAS_COV)
    assert np.allclose(pre, ref, rtol=1e-10, atol=1e-13)
    assert pre[141] == pytest.approx(S * 0.005)
