"""CPU tests of the oracle's marginalization (IMU/marginalization.cpp:291-384 with the B-Q3 fix) and
of the prior edge (EdgeMarginalization, IMU/g2otypes.cpp:1423-1497)."""
import numpy as np
import pytest


def _window_with_first_kf_tracks(pkg, seed):
    w = pkg.window.make_window(12, 260, 50, imu=True, seed=seed)
    return w


def _independent_marg(orc, pkg, w, state, first_kf=0, NUM=50, old_prior=None):
    """numpy restatement of the factor selection + dense Schur with eigen pseudo-inverse, from the
    per-edge evaluators; returns (A', b', kept ids/sizes)."""
    cam = orc.cam_vec(w["cam"])
    kf = w["kf"]
    K = len(kf["P"])
    navs = [orc.nav_vec(state["P"][k], state["V"][k], state["q"][k], kf["bg"][k], kf["ba"][k], state["dbg"][k], state["dba"][k]) for k in range(K)]
    factors = []   # list of (r, [(pid, J)])
    im = w["imu"]
    i, j = im["kf_i"][0], im["kf_j"][0]
    e, J0, J1, J2 = orc.eval_pvr_edge(w["gw"], navs[i], navs[j], navs[i], im["preint"][0])
    factors.append((e, [(kf["vid_pvr"][i], J0), (kf["vid_pvr"][j], J1), (kf["vid_bias"][i], J2)]))
    eb = np.concatenate([(navs[j][10:13] + navs[j][16:19]) - (navs[i][10:13] + navs[i][16:19]),
                         (navs[j][13:16] + navs[j][19:22]) - (navs[i][13:16] + navs[i][19:22])])
    factors.append((eb, [(kf["vid_bias"][i], -np.eye(6)), (kf["vid_bias"][j], np.eye(6))]))
    drop = {int(kf["vid_pvr"][first_kf]), int(kf["vid_bias"][first_kf])}
    for (lm_of, kf_of, meas, lms, ev, base) in ((w["po_pt"], w["po_kf"], w["po_uv"], state["points"], orc.eval_point_edge, 1 << 28),
                                                 (w["lo_ln"], w["lo_kf"], w["lo_l"], state["lines"], orc.eval_line_edge, 1 << 29)):
        num = 0
        first_of = {}
        for e_i, l in enumerate(lm_of):
            first_of.setdefault(int(l), int(kf_of[e_i]))
        for e_i, l in enumerate(lm_of):
            if first_of[int(l)] != first_kf:
                continue
            k = kf_of[e_i]
            e, Ji, Jj, _ = ev(cam, navs[k], lms[l], meas[e_i])
            factors.append((e, [(base + int(l), Ji), (int(kf["vid_pvr"][k]), Jj)]))
            drop.add(base + int(l))
            num += 1
            if num > NUM:
                break
    if old_prior is not None:
        n0 = old_prior["n"]
        dx = np.zeros(n0)
        for v, s, ix, in zip(old_prior["vid"], old_prior["size"], old_prior["idx"]):
            k = (v // 2) - kf["vid_pvr"][0] // 2
        raise NotImplementedError
    ids = sorted({int(pid) for _, bl in factors for pid, _ in bl})
    sizes = {}
    for _, bl in factors:
        for pid, J in bl:
            sizes[int(pid)] = J.shape[1]
    order = [i_ for i_ in ids if i_ in drop] + [i_ for i_ in ids if i_ not in drop]
    off, pos = {}, 0
    for pid in order:
        off[pid] = pos; pos += sizes[pid]
    m = sum(sizes[i_] for i_ in ids if i_ in drop)
    A = np.zeros((pos, pos)); b = np.zeros(pos)
    for r, bl in factors:
        for pa, Ja in bl:
            b[off[int(pa)]:off[int(pa)] + Ja.shape[1]] += Ja.T @ r
            for pb, Jb in bl:
                A[off[int(pa)]:off[int(pa)] + Ja.shape[1], off[int(pb)]:off[int(pb)] + Jb.shape[1]] += Ja.T @ Jb
    Amm = 0.5 * (A[:m, :m] + A[:m, :m].T)
    wv, V = np.linalg.eigh(Amm)
    inv = V @ np.diag(np.where(wv > 1e-8, 1.0 / np.where(wv > 1e-8, wv, 1), 0)) @ V.T
    Ar = A[m:, m:] - A[m:, :m] @ inv @ A[:m, m:]
    br = b[m:] - A[m:, :m] @ inv @ b[:m]
    kept = [i_ for i_ in ids if i_ not in drop]
    return Ar, br, kept, [sizes[i_] for i_ in kept], m


def test_marginalization_matches_independent_dense_schur(orc, pkg):
    w = _window_with_first_kf_tracks(pkg, 21)
    p = orc.new_problem(); p.upload_window(w)
    pkg.protocol.local_ba(p)
    state = pkg.protocol.results(p)
    pr = p.marginalize(0, 50)
    Ar, br, kept, sizes, m = _independent_marg(orc, pkg, w, state)
    assert list(pr["vid"]) == kept and list(pr["size"]) == sizes and pr["m"] == m
    assert pr["n"] == sum(sizes)
    sc = np.abs(Ar).max()
    assert np.allclose(pr["Ar"], Ar, rtol=1e-6, atol=1e-9 * sc)
    assert np.allclose(pr["br"], br, rtol=1e-6, atol=1e-9 * np.abs(br).max())
    # J0^T J0 == A' with eigenvalues <= eps dropped; J0^T r0 == projection of b' on the kept eigenspace
    w2, V2 = np.linalg.eigh(pr["Ar"])
    keep = w2 > 1e-8
    Ath = (V2[:, keep] * w2[keep]) @ V2[:, keep].T
    J0 = pr["J0"]
    assert np.allclose(J0.T @ J0, Ath, rtol=1e-7, atol=1e-9 * sc)
    assert np.allclose(J0.T @ pr["r0"], V2[:, keep] @ (V2[:, keep].T @ pr["br"]), rtol=1e-6, atol=1e-8 * np.abs(br).max())
    # kept vertices: PVR -> 10 doubles (P, V, quat), bias -> 6; x0 is the final estimate (mapHandler.cpp:6087..6183)
    o = 0
    for v, s in zip(pr["vid"], pr["size"]):
        k = int(v) // 2
        if s == 9:
            assert np.allclose(pr["x0"][o:o + 3], state["P"][k]) and np.allclose(pr["x0"][o + 3:o + 6], state["V"][k])
            R = orc.quat_to_R(state["q"][k])
            assert np.allclose(orc.quat_to_R(pr["x0"][o + 6:o + 10]), R, atol=1e-12)
            o += 10
        else:
            assert np.allclose(pr["x0"][o:o + 3], w["kf"]["bg"][k] + state["dbg"][k]); o += 6
    # the cap: NUM=50 admits 51 edges of each kind (B-Q10)
    p.close()


def test_prior_edge_error_and_chained_marginalization(orc, pkg):
    """Window 0 -> prior; window 1 (shifted by one keyframe, first KF fixed) carries the prior edge:
    at the linearisation point the prior error is r0, it contributes J0^T J0 to the free kept vertices,
    and a second marginalization consumes the old prior as a factor."""
    w0 = pkg.window.make_window(12, 260, 50, imu=True, seed=22)
    p = orc.new_problem(); p.upload_window(w0)
    pkg.protocol.local_ba(p)
    s0 = pkg.protocol.results(p)
    pr = p.marginalize(0, 50)
    p.close()
    # next window = keyframes 1..11 of the same trajectory (+ their landmarks), estimates carried over
    w1 = pkg.window.make_window(12, 260, 50, imu=True, seed=22)
    keep_kf = np.arange(1, 12)
    kf = {k: (v[keep_kf] if hasattr(v, "shape") and len(v) == 12 else v) for k, v in w1["kf"].items()}
    kf["P"], kf["V"], kf["q"] = s0["P"][1:], s0["V"][1:], s0["q"][1:]
    kf["dbg"], kf["dba"] = s0["dbg"][1:], s0["dba"][1:]
    kf["fixed_pvr"] = np.zeros(11, np.uint8); kf["fixed_pvr"][0] = 1
    kf["fixed_bias"] = kf["fixed_pvr"].copy()
    w1["kf"] = kf
    selp = w1["po_kf"] >= 1
    first = {}
    for l, k in zip(w1["po_pt"], w1["po_kf"]):
        first.setdefault(int(l), int(k))
    loc = np.array([first[int(l)] >= 1 for l in w1["po_pt"]])     # local map: first obs >= first window KF (:5768-5781)
    selp &= loc
    remap = {l: i for i, l in enumerate(sorted(set(w1["po_pt"][selp].tolist())))}
    w1["points"] = s0["points"][sorted(remap)]
    w1["po_pt"] = np.array([remap[int(l)] for l in w1["po_pt"][selp]], np.int32)
    w1["po_kf"] = (w1["po_kf"][selp] - 1).astype(np.int32); w1["po_uv"] = w1["po_uv"][selp]; w1["po_w"] = w1["po_w"][selp]
    w1["lines"] = np.zeros((0, 6)); w1["lo_ln"] = np.zeros(0, np.int32); w1["lo_kf"] = np.zeros(0, np.int32)
    w1["lo_l"] = np.zeros((0, 3)); w1["lo_w"] = np.zeros(0)
    im = w1["imu"]
    w1["imu"] = dict(kf_i=(im["kf_i"][1:] - 1).astype(np.int32), kf_j=(im["kf_j"][1:] - 1).astype(np.int32),
                     preint=im["preint"][1:], info_pvr=im["info_pvr"][1:], info_bias=im["info_bias"][1:])
    w1["prior"] = pr
    assert pkg.protocol.next_window_prior_ok(pr, kf["vid_pvr"], kf["vid_bias"])
    q = orc.new_problem(); q.upload_window(w1)
    q.recompute_errors()
    chi, _ = q.edge_chi2(pkg.abi.EDGE_PRIOR)
    # dx = 0 at the linearisation point -> e = r0 (2*vec(q0^-1 q) == 0 up to rounding)
    q.debug_build(1.0, False)
    assert np.allclose(q.debug_get("err_prior"), pr["r0"], atol=1e-9 * max(1, np.abs(pr["r0"]).max()))
    assert chi[0] == pytest.approx(pr["r0"] @ pr["r0"], rel=1e-9)
    st = q.optimize(3)
    assert st.iterations == 3 and np.isfinite(st.chi2_final) and st.chi2_final <= st.chi2_initial
    pr2 = q.marginalize(0, 50)
    assert pr2["n"] > 0 and np.all(np.isfinite(pr2["J0"]))
    # vertices of the dropped keyframe are gone, survivors keep their ids
    assert int(kf["vid_pvr"][0]) not in pr2["vid"] and int(kf["vid_bias"][0]) not in pr2["vid"]
    assert set(int(v) for v in pr["vid"]) - {int(kf["vid_pvr"][0]), int(kf["vid_bias"][0])} <= set(int(v) for v in pr2["vid"])
    q.close()


def test_there_is_no_whitening_option(pkg, orc):
    """SURVEY B-Q4: the factors enter the prior unweighted, as IMU/marginalization.cpp:67 has them.  Round 3 declared a
    `whiten_marg_factors` option and rejected it at run time; it is gone from include/plba.h (an option that cannot be set cannot be
    silently ignored)."""
    with pytest.raises(pkg.abi.PlbaError, match="unknown option"):
        orc.new_problem(whiten_marg_factors=1)


def test_oracle_against_the_extended_precision_marginalization(pkg, orc):
    """tests/golden/marg_exact.npz holds A', b', r0^T r0 of five windows evaluated at 40 digits (make_marg_exact.py).  Where no
    eigenvalue of the dropped block Amm lies near the 1e-8 threshold the fp64 oracle reproduces them to 1e-10.  On the far-landmark
    windows (eigenvalues of Amm between 1e-9 and 1e-7 while |Amm| ~ 1e7) it is only good to ~1e-4: like the reference it forms
    Amm = J^T J in fp64, which already moves those eigenvalues — the documented limit of this oracle as an arbiter there
    (the HIP path is held to the 40-digit values instead, tests/test_gpu_parity.py)."""
    import os, sys
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, here)
    import marg_cases
    gold = np.load(os.path.join(here, "marg_exact.npz"))
    for name, spec in marg_cases.CASES:
        w = marg_cases.case_window(pkg, spec)
        o = orc.new_problem(); o.upload_window(w)
        po = o.marginalize(0, 50); o.close()
        Ar, br = gold[name + "_Ar"], gold[name + "_br"]
        assert (po["m"], po["n"]) == tuple(int(x) for x in gold[name + "_dims"][:2]) and list(po["vid"]) == list(gold[name + "_vid"])
        dA = np.abs(po["Ar"] - Ar).max() / np.abs(Ar).max()
        db = np.abs(po["br"] - br).max() / max(np.abs(br).max(), 1.0)
        far = spec.get("far", 1.0)
        assert dA < (1e-3 if far >= 1e2 else 1e-10) and db < (2e-3 if far >= 1e2 else 1e-10), (name, dA, db)
        r0r0 = float(gold[name + "_r0r0"][0])
        assert abs(po["r0"] @ po["r0"] - r0r0) / r0r0 < (3e-2 if far >= 1e2 else 1e-3), name
