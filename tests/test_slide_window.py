"""plba_slide_window (include/plba.h): the window that drops its oldest keyframe and gains one keeps its device-resident estimates and is
edited in place; the structure the next plba_optimize builds — and so every result, BIT FOR BIT — must be that of a fresh handle given the
same window through plba_set_* (VERDICT r04 item 1).  The reference's own sequence of calls: src/mapHandler.cpp:1178-1221 (one BA per new
keyframe), :4815-4825 (addKeyframeToSW / deleteKeyframeInSW), :5769-5783 (local map = landmarks first seen inside the window)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _after_upload(pkg, p, w, prior):
    for kind, d in w["huber"].items():      # (gating switched the point / line kernels off: a new graph has them again, mapHandler.cpp:5937)
        p.set_robust(kind, True, d)
    p.set_prior(prior)


def _ba(pkg, p, marg):
    r = pkg.protocol.local_ba(p)
    prior = p.marginalize(0, pkg.protocol.MARG_NUM) if marg else None
    res = pkg.protocol.results(p)
    st = (r["stage1"].iterations, r["stage1"].trials, r["stage1"].chi2_initial, r["stage1"].chi2_final, r["gated"], r["stage2"].iterations, r["stage2"].trials, r["stage2"].chi2_final,
          r["stage2"].lambda_final)
    return st, res, prior


def _same(a, b, what):
    assert a[0] == b[0], (what, a[0], b[0])
    for k in a[1]:
        assert np.array_equal(np.asarray(a[1][k]), np.asarray(b[1][k])), (what, k, np.abs(np.asarray(a[1][k]) - np.asarray(b[1][k])).max())
    if a[2] is not None or b[2] is not None:
        for k in ("n", "m"):
            assert a[2][k] == b[2][k], (what, k)
        for k in ("vid", "size", "idx", "x0", "J0", "r0", "Ar", "br"):
            assert np.array_equal(a[2][k], b[2][k]), (what, "prior", k)


@pytest.mark.parametrize("K,Np,Nl,marg,opts", [(12, 300, 60, False, {}), (12, 300, 60, True, {}), (12, 900, 200, True, dict(lm_fused=2)), (20, 2500, 500, True, dict(lm_fused=2)),
                                              (8, 150, 40, False, dict(lm_fused=0))])
def test_slid_windows_equal_fresh_uploads_bit_for_bit(pkg, hip, K, Np, Nl, marg, opts):
    W = pkg.window
    nwin = 4
    seq = W.make_sequence(K, nwin, Np, Nl, seed=0x511DE + K, kf_dt=0.1 if K <= 12 else 0.25)
    slid = pkg.new_problem(**opts)
    w_prev, res_prev, prior = None, None, None
    for i in range(nwin):
        w = W.window_at(seq, i, K, prev=w_prev)
        if i == 0:
            wf = w
            slid.upload_window(w)
        else:
            wf = W.window_from_results(w, w_prev, res_prev)
            pm, lm = slid.slide_window(W.slide_delta(w_prev, w))
            # the maps say where the previous window's landmarks went: exactly the ones the next window lists first
            assert np.array_equal(np.flatnonzero(pm >= 0), np.flatnonzero(np.isin(w_prev["ids"]["points"], w["ids"]["points"])))
            assert (slid.dims["Np"], slid.dims["Nl"], slid.dims["Ep"], slid.dims["El"]) == (len(w["points"]), len(w["lines"]), len(w["po_pt"]), len(w["lo_ln"]))
        _after_upload(pkg, slid, w, prior)
        fresh = pkg.new_problem(**opts)
        wf = dict(wf); wf["prior"] = prior
        fresh.upload_window(wf)
        a = _ba(pkg, slid, marg)
        b = _ba(pkg, fresh, marg)
        _same(a, b, "window %d" % i)
        fresh.close()
        w_prev, res_prev, prior = w, a[1], a[2]
    slid.close()


def test_slide_at_the_headline_shape_and_with_drop_masks(pkg, hip):
    """half of BASELINE configs[2] (the fused landmark passes and the multi-chain factorisation run), one slide; then a slide that also drops
    observations by mask (the culling between two BA calls, mapHandler.cpp:5541-5620) and a landmark by mask (removeBadMapLandmarks)"""
    W = pkg.window
    K = 50
    seq = W.make_sequence(K, 3, 10000, 2000, seed=0x511DE50)
    w0 = W.window_at(seq, 0, K)
    slid = pkg.new_problem(); slid.upload_window(w0)
    a0 = _ba(pkg, slid, False)
    assert int(slid.debug_get("lm_fused")[0]) == 1
    # slide 1: plain
    w1 = W.window_at(seq, 1, K, prev=w0)
    slid.slide_window(W.slide_delta(w0, w1)); _after_upload(pkg, slid, w1, None)
    fresh = pkg.new_problem(); fresh.upload_window(W.window_from_results(w1, w0, a0[1]))
    a1, b1 = _ba(pkg, slid, False), _ba(pkg, fresh, False)
    _same(a1, b1, "slide 1"); fresh.close()
    # slide 2: the culling decision of the previous BA drops observations, and one landmark is removed outright
    cull = slid.cull_observations(W.CHI2_GATE)
    w2 = W.window_at(seq, 2, K, prev=w1)
    d = W.slide_delta(w1, w2)
    bad_p, bad_l = cull["bad_points"].astype(np.uint8), cull["bad_lines"].astype(np.uint8)
    # (an observation is only dropped when its landmark keeps at least two, so that the fresh window below stays well-formed)
    def thin(bad, ob_lm):
        left = np.bincount(ob_lm[bad == 0], minlength=ob_lm.max() + 1)
        bad = bad.copy(); bad[left[ob_lm] < 2] = 0
        return bad
    bad_p, bad_l = thin(bad_p, w1["po_pt"]), thin(bad_l, w1["lo_ln"])
    assert bad_p.sum() > 10
    victim = int(np.flatnonzero(np.isin(w1["ids"]["points"], w2["ids"]["points"]))[5])      # a point that would stay
    # the slide's added observations must not refer to the victim
    keep_add = d["po_pt"] != victim
    for k in ("po_pt", "po_kf", "po_uv", "po_w"):
        d[k] = d[k][keep_add]
    dp = np.zeros(len(w1["points"]), np.uint8); dp[victim] = 1
    d["drop_point"], d["drop_point_obs"], d["drop_line_obs"] = dp, bad_p, bad_l
    pm, lm = slid.slide_window(d); _after_upload(pkg, slid, w2, None)
    assert pm[victim] == -1
    # the same window for a fresh handle: w2 from the results, minus the victim and the culled observations
    wf = W.window_from_results(w2, w1, a1[1])
    vic2 = int(np.flatnonzero(w2["ids"]["points"] == w1["ids"]["points"][victim])[0])
    old_obs_p = {int(o): j for j, o in enumerate(w1["ids"]["points_obs"])}
    old_obs_l = {int(o): j for j, o in enumerate(w1["ids"]["lines_obs"])}
    keep_p = np.array([(wf["po_pt"][j] != vic2) and not (int(o) in old_obs_p and bad_p[old_obs_p[int(o)]]) for j, o in enumerate(w2["ids"]["points_obs"])], bool)
    keep_l = np.array([not (int(o) in old_obs_l and bad_l[old_obs_l[int(o)]]) for o in w2["ids"]["lines_obs"]], bool)
    renum = np.cumsum(np.arange(len(wf["points"])) != vic2) - 1
    wf["points"] = np.delete(wf["points"], vic2, axis=0)
    wf["po_pt"] = renum[wf["po_pt"][keep_p]].astype(np.int32); wf["po_kf"] = wf["po_kf"][keep_p]; wf["po_uv"] = wf["po_uv"][keep_p]; wf["po_w"] = wf["po_w"][keep_p]
    wf["lo_ln"] = wf["lo_ln"][keep_l]; wf["lo_kf"] = wf["lo_kf"][keep_l]; wf["lo_l"] = wf["lo_l"][keep_l]; wf["lo_w"] = wf["lo_w"][keep_l]
    fresh = pkg.new_problem(); fresh.upload_window(wf)
    a2, b2 = _ba(pkg, slid, False), _ba(pkg, fresh, False)
    _same(a2, b2, "slide 2 (drop masks)")
    fresh.close(); slid.close()


def test_slide_refusals_leave_the_window_as_it_was(pkg, hip):
    W = pkg.window
    seq = W.make_sequence(8, 2, 150, 40, seed=0x511DE08, kf_dt=0.1)
    w0 = W.window_at(seq, 0, 8); w1 = W.window_at(seq, 1, 8, prev=w0)
    d = W.slide_delta(w0, w1)
    p = pkg.new_problem()
    with pytest.raises(pkg.abi.PlbaError, match="no window is resident"):
        p.slide_window(d)
    p.upload_window(w0)
    with pytest.raises(pkg.abi.PlbaError, match="no window is resident"):      # uploaded but never built on the device
        p.slide_window(d)
    s0 = p.optimize(2)
    gone = int(w0["po_pt"][np.flatnonzero(w0["po_kf"] == 0)[0]])      # a point seen from the leaving keyframe
    bad = dict(d); bad["po_pt"] = np.concatenate([[gone], d["po_pt"]]).astype(np.int32); bad["po_kf"] = np.concatenate([[3], d["po_kf"]]).astype(np.int32)
    bad["po_uv"] = np.concatenate([[[1.0, 2.0]], d["po_uv"]]); bad["po_w"] = np.concatenate([[1.0], d["po_w"]])
    o = np.argsort(bad["po_pt"], kind="stable")
    for k in ("po_pt", "po_kf", "po_uv", "po_w"):
        bad[k] = bad[k][o]
    with pytest.raises(pkg.abi.PlbaError, match="leaves the window"):
        p.slide_window(bad)
    dup = dict(d); j = int(np.flatnonzero(d["po_pt"] < len(w0["points"]))[0])      # a second observation of a kept point from a keyframe that already sees it
    kf_seen = int(w0["po_kf"][np.flatnonzero(w0["po_pt"] == d["po_pt"][j])[-1]]) - 1
    dup["po_kf"] = d["po_kf"].copy(); dup["po_kf"][j] = kf_seen
    with pytest.raises(pkg.abi.PlbaError, match="observed twice"):
        p.slide_window(dup)
    # the handle still holds window 0, untouched: the same iterations give the same numbers as a second fresh handle
    q = pkg.new_problem(); q.upload_window(w0); q.optimize(2)
    s1, t1 = p.optimize(3), q.optimize(3)
    assert (s1.chi2_final, s1.trials) == (t1.chi2_final, t1.trials)
    p.slide_window(d)      # and the valid slide still goes through
    p.close(); q.close()


def test_slide_at_a_200_keyframe_window(pkg, hip):
    """BASELINE configs[4]'s keyframe count (the four-chain nested factorisation plan, 43 dense tiles) at a third of its landmarks (0.35 M
    observations: fused passes, landmark estimates stored in group order): two consecutive slides, each bit-identical to a fresh handle
    given the same window — estimates, chi2 trace and the marginalization prior the next window carries."""
    W = pkg.window
    K, nwin = 200, 3
    seq = W.make_sequence(K, nwin, 66000, 13000, seed=0x511DE200, kf_dt=0.05)
    slid = pkg.new_problem()
    w_prev, res_prev, prior = None, None, None
    for i in range(nwin):
        w = W.window_at(seq, i, K, prev=w_prev)
        if i == 0:
            wf = w
            slid.upload_window(w)
        else:
            wf = W.window_from_results(w, w_prev, res_prev)
            slid.slide_window(W.slide_delta(w_prev, w))
        _after_upload(pkg, slid, w, prior)
        a = _ba(pkg, slid, True)
        if i > 0:
            fresh = pkg.new_problem()
            wf = dict(wf); wf["prior"] = prior
            fresh.upload_window(wf)
            b = _ba(pkg, fresh, True)
            _same(a, b, "window %d" % i)
            fresh.close()
        w_prev, res_prev, prior = w, a[1], a[2]
    assert int(slid.debug_get("lm_fused")[0]) == 1 and int(slid.debug_get("twin")[0]) == 1
    slid.close()
