"""the pre-VIO-init visual-only local BA, MapHandler::levMarquardtOptimizationLBA
(src/mapHandler.cpp:1441-2098) — plba_lba_visual against the oracle's restatement orc_lba_visual, and the oracle against
independent numpy checks (the reference holds no fixture for this function either: parity unpinned)."""
import numpy as np
import pytest


def _run(prob, w, **opts):
    return prob.lba_visual(w["T_kf_w"], w["kf_loc"], w["xyz"], w["pq"], w["po_pt"], w["po_kf"], w["uv"], w["lo_ln"], w["lo_kf"], w["l3"], w["cam"], **opts)


def _se3_exp(x):      # stvo-pl/src/auxiliar.cpp:124-141, numpy
    t, w = x[:3], x[3:]
    th = np.linalg.norm(w)
    T = np.eye(4)
    if th < 1e-6:
        T[:3, 3] = t
        return T
    s = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / th
    T[:3, :3] = np.eye(3) + s * np.sin(th) + s @ s * (1 - np.cos(th))
    V = np.eye(3) + s * (1 - np.cos(th)) / th + s @ s * (th - np.sin(th)) / th
    T[:3, 3] = V @ t
    return T


def _point_residuals(w, T, xyz):
    fx, fy, cx, cy = w["cam"]
    r = np.zeros(len(w["po_pt"]))
    for e, (i, k) in enumerate(zip(w["po_pt"], w["po_kf"])):
        Xc = T[k][:3, :3].T @ (xyz[i] - T[k][:3, 3])
        r[e] = np.hypot(w["uv"][e, 0] - (cx + fx * Xc[0] / Xc[2]), w["uv"][e, 1] - (cy + fy * Xc[1] / Xc[2]))
    return r


def test_oracle_first_step_is_the_gauss_newton_step_of_the_norm_residuals(pkg, orc):
    """points only, one pass: X1 must be the damped, weighted Gauss-Newton step on r_e = |e_e| (weights 1 / (1 + r^2) frozen;
    a norm residual gives ONE equation per observation, so the undamped system is rank deficient for short tracks and the
    multiplicative damping lambda = lambdaLbaLM max |H_ii| is part of the step) computed from NUMERIC derivatives under the
    reference's own update rule T <- T expmap(dx)^-1 — pins the analytic pose / landmark Jacobian rows (:1490-1512), the
    assembly, the lambda initialisation (:1653-1663), the solve and the update (:1670-1680) in one go"""
    w = pkg.window.make_visual_window(K=5, Np=40, Nl=0, n_fixed=2, seed=11)
    o = orc.new_problem()
    res = _run(o, w, max_iters=1)
    assert res["iterations"] == 1 and res["updates"] == 1
    loc = w["kf_loc"]; Nkf = int((loc >= 0).sum()); Np = len(w["xyz"]); N = 6 * Nkf + 3 * Np

    def resid(dx):
        T = w["T_kf_w"].copy()
        for k in range(len(loc)):
            if loc[k] >= 0:
                T[k] = T[k] @ np.linalg.inv(_se3_exp(dx[6 * loc[k]:6 * loc[k] + 6]))
        return _point_residuals(w, T, w["xyz"] + dx[6 * Nkf:].reshape(-1, 3))
    r0 = resid(np.zeros(N))
    J = np.zeros((len(r0), N))
    h = 1e-6
    for c in range(N):
        d = np.zeros(N); d[c] = h
        J[:, c] = (resid(d) - resid(-d)) / (2 * h)
    W = 1.0 / (1.0 + r0 ** 2)
    H = J.T @ (W[:, None] * J)
    lam = 1e-5 * np.abs(np.diag(H)).max()
    assert res["lam"] == pytest.approx(lam, rel=1e-6)
    dx = np.linalg.solve(H + lam * np.diag(np.diag(H)), -J.T @ (W * r0))
    T1 = w["T_kf_w"].copy()
    for k in range(len(loc)):
        if loc[k] >= 0:
            T1[k] = T1[k] @ np.linalg.inv(_se3_exp(dx[6 * loc[k]:6 * loc[k] + 6]))
    assert np.abs(res["xyz"] - (w["xyz"] + dx[6 * Nkf:].reshape(-1, 3))).max() < 2e-6
    assert np.abs(res["T"] - T1).max() < 2e-6
    assert res["err_first"] == pytest.approx((W * r0 ** 2).sum() / len(r0), rel=1e-12)
    o.close()


def test_oracle_run_follows_the_coded_schedule(pkg, orc):
    w = pkg.window.make_visual_window(K=6, Np=120, Nl=30, n_fixed=2, seed=5)
    o = orc.new_problem()
    res = _run(o, w)
    assert 1 <= res["iterations"] <= 15 and res["updates"] >= 1 and not res["solver_failed"]
    # lambda starts at lambdaLbaLM * max |H_ii| and moves by factors of lambdaLbaK only (:1653-1659, :1894-1900)
    one = _run(o, w, max_iters=1)
    ratio = res["lam"] / one["lam"]
    assert np.log10(ratio) == pytest.approx(round(np.log10(ratio)), abs=1e-9)
    # the first pass' error is x / 0 = +inf as coded (:1650), so the first comparison always counts as a success
    assert np.isinf(one["err_last"]) and _run(o, w, max_iters=2)["updates"] == 2
    # a step is only taken when the (landmark-normalised, :1882) error did not grow: the sequence of pass errors is monotone
    errs = [_run(o, w, max_iters=it)["err_last"] for it in range(2, res["iterations"] + 1)]
    assert all(b <= a for a, b in zip(errs[:-1], errs[1:]))
    # fixed keyframes come back untouched
    fixed = w["kf_loc"] < 0
    assert np.array_equal(res["T"][fixed], w["T_kf_w"][fixed])
    # the moved flags are the 1 cm test of :1944-1970
    assert np.array_equal(res["pt_moved"], np.linalg.norm(res["xyz"] - w["xyz"], axis=1) > 0.01)
    assert np.array_equal(res["ln_moved"], np.linalg.norm(res["pq"] - w["pq"], axis=1) > 0.01)
    # the line pass linearises at the map poses (:1790) unless told otherwise: the two runs must differ once lines are in
    alt = _run(o, w, use_iterate_poses=1)
    assert np.abs(alt["T"] - res["T"]).max() > 1e-9
    o.close()


def test_oracle_gba_variant_takes_every_step(pkg, orc):
    """levMarquardtOptimizationGBA (:2210-2812): err is x / 0 in every pass (:2744), so no step is ever refused, lambda
    only grows, and lambda starts from the TRUNCATED max |H_ii| (`int Hmax`, :2468)"""
    w = pkg.window.make_visual_window(K=6, Np=100, Nl=20, n_fixed=1, seed=8)
    o = orc.new_problem()
    eps = float(np.finfo(float).eps)
    r = _run(o, w, variant=1, min_error=eps, min_error_change=eps, max_iters=6)
    assert r["iterations"] == 6 and r["updates"] == 6 and np.isinf(r["err_last"])
    l0 = _run(o, w, variant=0, max_iters=1)["lam"]
    l1 = _run(o, w, variant=1, max_iters=1)["lam"]
    assert l1 == pytest.approx(1e-5 * np.floor(l0 / 1e-5), rel=1e-12) and l1 < l0
    assert r["lam"] == pytest.approx(l1 * 10.0 ** 5, rel=1e-12)
    o.close()


@pytest.mark.gpu
def test_device_gba_variant_matches_the_oracle(pkg, orc, hip):
    w = pkg.window.make_visual_window(K=7, Np=200, Nl=40, n_fixed=1, seed=13)
    eps = float(np.finfo(float).eps)
    g = pkg.new_problem(); o = orc.new_problem()
    a, b = (_run(x, w, variant=1, min_error=eps, min_error_change=eps, max_iters=8) for x in (g, o))
    assert (a["iterations"], a["updates"]) == (b["iterations"], b["updates"]) == (8, 8)
    assert a["lam"] == pytest.approx(b["lam"], rel=1e-12)
    assert np.abs(a["T"] - b["T"]).max() < 1e-9 and np.abs(a["xyz"] - b["xyz"]).max() < 1e-8 and np.abs(a["pq"] - b["pq"]).max() < 1e-8
    g.close(); o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("K,Np,Nl,nfix,seed,iterate", [(6, 150, 30, 2, 1, 0), (8, 300, 0, 1, 2, 0), (5, 0, 60, 2, 3, 0), (10, 400, 80, 3, 4, 1), (3, 60, 12, 1, 5, 0)])
def test_device_lba_matches_the_oracle(pkg, orc, hip, K, Np, Nl, nfix, seed, iterate):
    w = pkg.window.make_visual_window(K=K, Np=Np, Nl=Nl, n_fixed=nfix, seed=seed)
    g = pkg.new_problem(); o = orc.new_problem()
    a, b = _run(g, w, use_iterate_poses=iterate), _run(o, w, use_iterate_poses=iterate)
    assert (a["iterations"], a["updates"], a["solver_failed"]) == (b["iterations"], b["updates"], 0)
    assert a["err_first"] == pytest.approx(b["err_first"], rel=1e-12)
    assert a["err_last"] == pytest.approx(b["err_last"], rel=1e-9)
    assert a["lam"] == pytest.approx(b["lam"], rel=1e-12)
    assert np.abs(a["T"] - b["T"]).max() < 1e-9
    assert np.abs(a["xyz"] - b["xyz"]).max() < 1e-8 if Np else True
    assert np.abs(a["pq"] - b["pq"]).max() < 1e-8 if Nl else True
    assert np.array_equal(a["pt_moved"], b["pt_moved"]) and np.array_equal(a["ln_moved"], b["ln_moved"])
    g.close(); o.close()


@pytest.mark.gpu
def test_device_lba_single_steps_match_the_oracle(pkg, orc, hip):
    """one pass and two passes separately: the first solve (lambda from max |H_ii|) and the first re-linearisation"""
    w = pkg.window.make_visual_window(K=6, Np=200, Nl=40, n_fixed=2, seed=9)
    g = pkg.new_problem(); o = orc.new_problem()
    for it in (1, 2, 3):
        a, b = _run(g, w, max_iters=it), _run(o, w, max_iters=it)
        assert (a["iterations"], a["updates"]) == (b["iterations"], b["updates"])
        assert a["lam"] == pytest.approx(b["lam"], rel=1e-12)
        assert np.abs(a["T"] - b["T"]).max() < 1e-10 and np.abs(a["xyz"] - b["xyz"]).max() < 1e-9 and np.abs(a["pq"] - b["pq"]).max() < 1e-9
    g.close(); o.close()


@pytest.mark.gpu
def test_device_lba_larger_window_properties(pkg, hip):
    """a window the oracle's dense LDL^T (N = 6 Nkf + 3 Np + 6 Nl) would need minutes for: properties only"""
    w = pkg.window.make_visual_window(K=20, Np=6000, Nl=1200, n_fixed=2, seed=21)
    g = pkg.new_problem()
    a = _run(g, w)
    assert not a["solver_failed"] and a["updates"] >= 1 and np.isfinite(a["T"]).all() and np.isfinite(a["xyz"]).all()
    fixed = w["kf_loc"] < 0
    assert np.array_equal(a["T"][fixed], w["T_kf_w"][fixed])
    errs = [_run(g, w, max_iters=it)["err_last"] for it in range(2, a["iterations"] + 1)]
    assert all(b <= a_ for a_, b in zip(errs[:-1], errs[1:]))
    b = _run(g, w)      # same input, same answer (the pose-system atomics add in any order: not bit-equal, but close)
    assert np.abs(a["T"] - b["T"]).max() < 1e-10
    g.close()


@pytest.mark.gpu
def test_device_lba_rejects_malformed_lists(pkg, hip):
    w = pkg.window.make_visual_window(K=4, Np=30, Nl=6, n_fixed=1, seed=2)
    g = pkg.new_problem()
    bad = dict(w); bad["po_pt"] = w["po_pt"][::-1].copy()
    with pytest.raises(pkg.abi.PlbaError, match="ordered by point"):
        _run(g, bad)
    bad = dict(w); bad["kf_loc"] = np.array([-1, 0, 0, 1], np.int32)
    with pytest.raises(pkg.abi.PlbaError, match="once each"):
        _run(g, bad)
    bad = dict(w); bad["po_kf"] = w["po_kf"].copy(); bad["po_kf"][0] = 9
    with pytest.raises(pkg.abi.PlbaError, match="out of range"):
        _run(g, bad)
    g.close()


@pytest.mark.gpu
def test_lba_call_site_through_the_harness(pkg, orc, hip, tmp_path):
    """tools/localba_harness.cpp `lba`: MapHandler::localBundleAdjustment's list building on map-shaped objects and
    levMarquardtOptimizationLBA's body as one plba_lba_visual call, write-back included, against the oracle"""
    import subprocess
    from .test_facade import build_harness
    exe = build_harness()
    w = pkg.window.make_visual_window(K=7, Np=160, Nl=30, n_fixed=2, seed=33)
    fin, fout = str(tmp_path / "lba_in.bin"), str(tmp_path / "lba_out.bin")
    K, Np, Nl, Ep, El = len(w["T_kf_w"]), len(w["xyz"]), len(w["pq"]), len(w["po_pt"]), len(w["lo_ln"])
    with open(fin, "wb") as f:
        np.array([K, Np, Nl, Ep, El], np.int32).tofile(f)
        np.array(w["cam"], np.float64).tofile(f); np.ascontiguousarray(w["T_kf_w"], np.float64).tofile(f); w["kf_loc"].astype(np.int32).tofile(f)
        np.ascontiguousarray(w["xyz"], np.float64).tofile(f); np.ascontiguousarray(w["pq"], np.float64).tofile(f)
        w["po_pt"].astype(np.int32).tofile(f); w["po_kf"].astype(np.int32).tofile(f); np.ascontiguousarray(w["uv"], np.float64).tofile(f)
        w["lo_ln"].astype(np.int32).tofile(f); w["lo_kf"].astype(np.int32).tofile(f); np.ascontiguousarray(w["l3"], np.float64).tofile(f)
    subprocess.check_call([exe, "lba", fin, fout])
    raw = np.fromfile(fout, np.uint8)
    its, ups = np.frombuffer(raw[:8].tobytes(), np.int32)
    off = 8
    T = np.frombuffer(raw[off:off + 128 * K].tobytes(), np.float64).reshape(K, 4, 4); off += 128 * K
    xyz = np.frombuffer(raw[off:off + 24 * Np].tobytes(), np.float64).reshape(Np, 3); off += 24 * Np
    pq = np.frombuffer(raw[off:off + 48 * Nl].tobytes(), np.float64).reshape(Nl, 6); off += 48 * Nl
    inl = np.frombuffer(raw[off:off + 4 * (Np + Nl)].tobytes(), np.int32)
    o = orc.new_problem(); b = _run(o, w); o.close()
    assert (its, ups) == (b["iterations"], b["updates"])
    assert np.abs(T - b["T"]).max() < 1e-9 and np.abs(xyz - b["xyz"]).max() < 1e-8 and np.abs(pq - b["pq"]).max() < 1e-8
    assert np.array_equal(inl[:Np] == 0, b["pt_moved"]) and np.array_equal(inl[Np:] == 0, b["ln_moved"])
