"""SURVEY §8f row 4 / Boundary 1: the OTHER g2o users of src/mapHandler.cpp run on the facade.

tools/localba_harness.cpp holds, next to the local-BA call sites, an IMUInitEstBg-shaped function (src/mapHandler.cpp:4989-5036:
one VertexGyrBias, EdgeGyrBias per keyframe pair, Levenberg, optimize(1)) and a pose-graph function in the shape of
loopClosureOptimizationCovGraphG2O (:4299-4528: BlockSolver<BlockSolverTraits<6,3>>, LinearSolverCholmod, VertexSE3 / EdgeSE3,
SE3Quat::exp, userLambdaInit 1e-10, computeInitialGuess, estimateAsSE3Quat).  Those graphs are host-evaluated: the facade runs
g2o's LM loop on the host (no GPU needed, so these tests run in the CPU suite); expected values come from the oracle's own
restatement (oracle/plba_oracle.c: lm_dense, orc_gyrbias_estimate, orc_pgo — EdgeSE3 Jacobians by central differences there,
analytic in the facade)."""
import ctypes as C
import subprocess

import numpy as np
import pytest

from .test_facade import build_harness


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _se3_exp(orc, x6):
    """SE3Quat::exp of (omega, upsilon) -> (R row-major 3x3, t), through the oracle's restatement of IMU/se3quat.h"""
    lib = orc.lib().cdll
    q, t, R = np.zeros(4), np.zeros(3), np.zeros(9)
    x6 = np.ascontiguousarray(x6, np.float64)
    lib.orc_se3_exp(_dp(x6), _dp(q), _dp(t))
    lib.orc_quat_to_R(_dp(q), _dp(R))
    return R.reshape(3, 3), t


def _pose12(R, t):
    return np.concatenate([R.ravel(), t])


def _gyr_problem(pkg, M=9, seed=5):
    """the inputs IMUInitEstBg assembles: per keyframe pair the preintegrated rotation, its bias Jacobian, the two orientations
    (vision-only estimates, here the truth) and the rotation block of the preintegration covariance; the IMU was integrated with
    a gyro bias that is off by `bias`, which the estimator has to find"""
    W = pkg.window
    rng = np.random.default_rng(seed)
    bias = np.array([0.012, -0.007, 0.009])
    S = int(round(W.KF_DT / W.IMU_DT))
    tk = W.KF_DT * np.arange(M + 1)
    ts = tk[:-1, None] + W.IMU_DT * (np.arange(S)[None, :] + 0.5)
    w_meas = W.traj_omega_body(ts) + bias + rng.normal(size=(M, S, 3)) * 1e-4
    pre = W.preintegrate(w_meas, np.zeros((M, S, 3)), W.IMU_DT)
    Rk = W.traj_R(tk)
    dR, JRg = pre[:, 6:15].copy(), pre[:, 51:60].copy()
    info = np.stack([np.linalg.inv(pre[m, 60:141].reshape(9, 9)[6:9, 6:9]) for m in range(M)]).reshape(M, 9)
    return dict(M=M, dR=dR, JRg=JRg, Ri=Rk[:-1].reshape(M, 9).copy(), Rj=Rk[1:].reshape(M, 9).copy(), info=info.copy(), bias=bias)


@pytest.mark.parametrize("iters", [1, 4])
def test_imu_init_est_bg_on_the_facade(pkg, orc, tmp_path, iters):
    g = _gyr_problem(pkg)
    exe = build_harness()
    fin, fout = str(tmp_path / "g.bin"), str(tmp_path / "g.out")
    with open(fin, "wb") as f:
        np.array([g["M"], iters], np.int32).tofile(f)
        for k in ("dR", "JRg", "Ri", "Rj", "info"):
            np.ascontiguousarray(g[k], np.float64).tofile(f)
    subprocess.check_call([exe, "gyrbias", fin, fout], timeout=60)
    r = np.fromfile(fout, np.float64)
    bg, st = np.zeros(3), np.zeros(4)
    a = {k: np.ascontiguousarray(g[k], np.float64) for k in ("dR", "JRg", "Ri", "Rj", "info")}
    orc.lib().cdll.orc_gyrbias_estimate(g["M"], _dp(a["dR"]), _dp(a["JRg"]), _dp(a["Ri"]), _dp(a["Rj"]), _dp(a["info"]), iters, 0, _dp(bg), _dp(st))
    assert np.abs(r[:3] - bg).max() < 1e-10 * max(1.0, np.abs(bg).max())
    assert r[3] == pytest.approx(st[0], rel=1e-10) and r[4] == pytest.approx(st[1], rel=1e-8, abs=1e-12) and int(r[5]) == int(st[2])
    # "it's actually a linear estimator, so 1 iteration is enough" (src/mapHandler.cpp:5027): the bias comes out after one step
    assert np.abs(r[:3] + g["bias"]).max() < 2e-3 or np.abs(r[:3] - g["bias"]).max() < 2e-3
    assert r[4] < 1e-2 * r[3]


def _pgo_problem(pkg, orc, nv=24, seed=11):
    """keyframes on the SURVEY §8d trajectory (world -> keyframe transforms, as the reference stores T_kf_w), odometry edges between
    neighbours and between keyframes two apart, drifted initial estimates, one loop closure from the last keyframe to the first"""
    W = pkg.window
    rng = np.random.default_rng(seed)
    tk = 0.6 * np.arange(nv)
    Rwb, Pwb = W.traj_R(tk), W.traj_p(tk)
    T = [np.block([[Rwb[k].T, (-Rwb[k].T @ Pwb[k])[:, None]], [np.zeros((1, 3)), np.ones((1, 1))]]) for k in range(nv)]

    def log6(Tm):                      # SE3Quat(R, t).log() = (omega, upsilon), through the oracle's restatement of IMU/se3quat.h:178-215
        lib = orc.lib().cdll
        R = np.ascontiguousarray(Tm[:3, :3]).ravel(); t = np.ascontiguousarray(Tm[:3, 3]); q = np.zeros(4); x = np.zeros(6)
        lib.orc_R_to_quat(_dp(R), _dp(q))
        q = q / np.linalg.norm(q) * (1.0 if q[3] >= 0 else -1.0)
        lib.orc_se3_log(_dp(q), _dp(t), _dp(x))
        return x
    edges = [(i, i + 1) for i in range(nv - 1)] + [(i, i + 2) for i in range(0, nv - 2, 3)]
    meas = []
    for i, j in edges:
        Z = np.linalg.inv(T[i]) @ T[j]
        Z[:3, :3] = Z[:3, :3] @ W.exp_so3(rng.normal(size=3) * 2e-3)
        Z[:3, 3] += rng.normal(size=3) * 5e-3
        meas.append(log6(Z))
    lc = [(0, nv - 1)]
    lc_meas = [log6(np.linalg.inv(T[0]) @ T[nv - 1])]
    est = []
    drift = np.eye(4)
    for k in range(nv):
        d = np.eye(4); d[:3, :3] = W.exp_so3(rng.normal(size=3) * 4e-3); d[:3, 3] = rng.normal(size=3) * 0.01
        drift = drift @ d
        est.append(log6(T[k] @ drift) if k else log6(T[0]))
    return dict(nv=nv, vid=np.arange(nv, dtype=np.int32) * 3 + 1, fixed=np.array([1] + [0] * (nv - 1), np.int32), est=np.array(est),
                edges=np.array(edges + lc, np.int32), meas=np.array(meas + lc_meas), ne=len(edges), nlc=len(lc))


def _run_pgo(pkg, orc, tmp_path, g, mode, init_guess, iters=6):
    exe = build_harness()
    fin, fout = str(tmp_path / "p.bin"), str(tmp_path / "p.out")
    with open(fin, "wb") as f:
        np.array([g["nv"], g["ne"], g["nlc"], iters, init_guess], np.int32).tofile(f)
        g["vid"].tofile(f); g["fixed"].tofile(f); np.ascontiguousarray(g["est"], np.float64).tofile(f)
        g["vid"][g["edges"][:, 0]].astype(np.int32).tofile(f); g["vid"][g["edges"][:, 1]].astype(np.int32).tofile(f)
        np.ascontiguousarray(g["meas"], np.float64).tofile(f)
    subprocess.check_call([exe, mode, fin, fout], timeout=300)
    r = np.fromfile(fout, np.float64)
    nv = g["nv"]
    got = np.array([_pose12(*_se3_exp(orc, r[6 * k: 6 * k + 6])) for k in range(nv)])
    # the oracle's restatement of the same protocol
    pose = np.array([_pose12(*_se3_exp(orc, x)) for x in g["est"]])
    meas = np.array([_pose12(*_se3_exp(orc, x)) for x in g["meas"]])
    ne = len(g["edges"])
    info = np.tile(np.eye(6).ravel(), (ne, 1))
    ei, ej = np.ascontiguousarray(g["edges"][:, 0], np.int32), np.ascontiguousarray(g["edges"][:, 1], np.int32)
    st = np.zeros(4)
    pose = np.ascontiguousarray(pose)
    orc.lib().cdll.orc_pgo(nv, _ip(g["fixed"]), _dp(pose), ne, _ip(ei), _ip(ej), _dp(np.ascontiguousarray(meas)), _dp(np.ascontiguousarray(info)),
                           iters, C.c_double(1e-10), 0, 1 if mode == "essgraph" else init_guess, _dp(st))
    return dict(got=got, pose=pose, chi0=r[6 * nv], chi1=r[6 * nv + 1], done=int(r[6 * nv + 2]), trials=int(r[6 * nv + 3]), device_solves=int(r[6 * nv + 4]), st=st)


def _check_small_pgo(pkg, orc, tmp_path, init_guess):
    g = _pgo_problem(pkg, orc)
    o = _run_pgo(pkg, orc, tmp_path, g, "pgo", init_guess)
    got, pose, st, chi0, chi1 = o["got"], o["pose"], o["st"], o["chi0"], o["chi1"]
    assert chi0 == pytest.approx(st[0], rel=1e-9)
    assert o["done"] == int(st[2]) and o["device_solves"] == 0      # 138 dims: the facade's host Cholesky
    if init_guess == 0:
        assert o["trials"] == int(st[3])                                  # same LM trial sequence
        assert chi1 == pytest.approx(st[1], rel=1e-5)                     # numeric (oracle) vs analytic (facade) Jacobians
        assert np.abs(got - pose).max() < 1e-6
        assert chi1 < 0.2 * chi0                                          # the loop closure pulled the drifted chain in
    else:
        # propagated start: the odometry is already satisfied, what is left (3e-3) is the loop closure's share; with lambda
        # = 1e-10 the last iterations accept / reject at rounding level, so the number of trials is not pinned
        assert chi1 == pytest.approx(st[1], rel=1e-3) and chi1 < chi0
        assert np.abs(got - pose).max() < 1e-5
    assert np.abs(got[0] - _pose12(*_se3_exp(orc, g["est"][0]))).max() < 1e-12      # the fixed vertex did not move


@pytest.mark.parametrize("init_guess", [0, 1])
def test_pose_graph_optimisation_on_the_facade(pkg, orc, tmp_path, init_guess):
    _check_small_pgo(pkg, orc, tmp_path, init_guess)


def _vio_problem(pkg, N=12, seed=3):
    """what tryVioInit has in hand (src/mapHandler.cpp:4827-4980): vision-only camera poses of N keyframes (here the truth), the
    preintegrated deltas between them — integrated with the true gyro bias removed but the accelerometer bias NOT (that is what the
    step estimates) — and the extrinsics"""
    W = pkg.window
    rng = np.random.default_rng(seed)
    S = int(round(W.KF_DT / W.IMU_DT))
    tk = W.KF_DT * np.arange(N)
    ts = tk[:-1, None] + W.IMU_DT * (np.arange(S)[None, :] + 0.5)
    Rs = W.traj_R(ts)
    w_meas = W.traj_omega_body(ts) + rng.normal(size=(N - 1, S, 3)) * 1e-5
    a_meas = np.einsum("msji,msj->msi", Rs, W.traj_a(ts) - W.GW) + W.BA_TRUE + rng.normal(size=(N - 1, S, 3)) * 1e-4
    pre = W.preintegrate(w_meas, a_meas, W.IMU_DT)
    Rwb, Pwb = W.traj_R(tk), W.traj_p(tk)
    Rbc, Pbc = W.T_BS[:3, :3], W.T_BS[:3, 3]
    Rc = np.einsum("kij,jl->kil", Rwb, Rbc); pc = Pwb + np.einsum("kij,j->ki", Rwb, Pbc)
    return dict(N=N, dt=pre[:, 141].copy(), dP=pre[:, 0:3].copy(), dV=pre[:, 3:6].copy(), JPa=pre[:, 24:33].copy(), JVa=pre[:, 42:51].copy(),
                Rc=Rc.reshape(N, 9).copy(), pc=pc.copy(), Rb=Rwb.reshape(N, 9).copy(), pb=Pwb.copy(), Rcb=Rbc.T.ravel().copy(), pcb=(-Rbc.T @ Pbc).copy(),
                V_true=W.traj_v(tk))


def test_vio_init_closed_form_steps(pkg, orc, tmp_path):
    """VERDICT r02 missing #3: tryVioInit's gravity and accelerometer-bias least-squares steps and the velocities
    (src/mapHandler.cpp:4853-4980) as include/plba_g2o/vio_init.h, re-enacted by the harness (`vioinit`), against the oracle's
    independent restatement (orc_vio_init: Householder QR instead of the normal equations' eigen-decomposition) and the generating truth."""
    g = _vio_problem(pkg)
    exe = build_harness()
    fin, fout = str(tmp_path / "v.bin"), str(tmp_path / "v.out")
    keys = ("dt", "dP", "dV", "JPa", "JVa", "Rc", "pc", "Rb", "pb", "Rcb", "pcb")
    with open(fin, "wb") as f:
        np.array([g["N"]], np.int32).tofile(f)
        for k in keys:
            np.ascontiguousarray(g[k], np.float64).tofile(f)
    subprocess.check_call([exe, "vioinit", fin, fout], timeout=60)
    r = np.fromfile(fout, np.float64)
    N = g["N"]
    a = {k: np.ascontiguousarray(g[k], np.float64) for k in keys}
    gpre, g0, ba, V = np.zeros(3), np.zeros(3), np.zeros(3), np.zeros(3 * N)
    f = orc.lib().cdll.orc_vio_init
    f.restype = None
    f.argtypes = [C.c_int] + [C.POINTER(C.c_double)] * 15
    f(N, *[_dp(a[k]) for k in ("dt", "dP", "dV", "JPa", "JVa", "Rc", "pc", "Rb", "pb", "Rcb", "pcb")], _dp(gpre), _dp(g0), _dp(ba), _dp(V))
    assert np.abs(r[0:3] - gpre).max() < 1e-9 * np.abs(gpre).max() and np.abs(r[3:6] - g0).max() < 1e-9 * 9.81
    assert np.abs(r[6:9] - ba).max() < 1e-8 and np.abs(r[9:] - V).max() < 1e-8
    # sanity against what generated the data (the gravity step ignores the accelerometer bias, so it is off by ~|ba| / g = 0.3 degrees,
    # and three seconds of this trajectory observe the bias only weakly: the reference's estimator, not a defect of the restatement)
    assert np.abs(r[3:6] - pkg.window.GW).max() < 0.1 and abs(np.linalg.norm(r[3:6]) - 9.81) < 1e-12
    assert np.abs(r[6:9] - pkg.window.BA_TRUE).max() < 0.06
    assert np.abs(r[9:].reshape(N, 3) - g["V_true"]).max() < 0.1


def _ess_graph_problem(pkg, orc, nv=80, seed=12):
    """the essential graph of loopClosureOptimizationEssGraphG2O (src/mapHandler.cpp:4068-4297): every keyframe from the first loop
    keyframe on, BOTH loop ends fixed (the corrected poses), spanning-tree edges, strong-covisibility edges, the loop edge: 78 free
    vertices = 468 dims, beyond what the facade solves on the host"""
    g = _pgo_problem(pkg, orc, nv=nv, seed=seed)
    g["fixed"] = np.array([1] + [0] * (nv - 2) + [1], np.int32)
    return g


@pytest.mark.gpu
def test_essential_graph_optimisation_uses_the_device_solver(pkg, orc, hip, tmp_path):
    """VERDICT r02 missing #3 / next #9: a pose graph of more than 384 dims goes through plba_debug_dense_solve (K7's kernels)"""
    g = _ess_graph_problem(pkg, orc)
    o = _run_pgo(pkg, orc, tmp_path, g, "essgraph", 1)
    assert o["device_solves"] >= o["done"] >= 1
    assert o["chi0"] == pytest.approx(o["st"][0], rel=1e-8)
    assert o["chi1"] == pytest.approx(o["st"][1], rel=1e-3) and o["chi1"] < o["chi0"]
    assert np.abs(o["got"] - o["pose"]).max() < 1e-5
    for k in (0, g["nv"] - 1):
        assert np.abs(o["got"][k] - _pose12(*_se3_exp(orc, g["est"][k]))).max() < 1e-12      # the loop ends stayed where they were put


@pytest.mark.gpu
def test_host_evaluated_graphs_on_the_gpu_box(pkg, orc, hip, tmp_path):
    """the same small graphs as the CPU suite, on the box the harness is meant to run on (it links libplba_hip.so)"""
    _check_small_pgo(pkg, orc, tmp_path, 0)
