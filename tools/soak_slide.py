"""Soak of plba_slide_window: random sequences (window length, landmark counts, track lengths incl. wide groups, revisits, with / without
marginalization priors, every landmark path), `nslides` consecutive slides each, every slid call compared BIT FOR BIT with a fresh handle
given the same window.  python tools/soak_slide.py [N] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); W = pkg.window
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261005)
bad = 0
for case in range(N):
    K = int(rng.integers(5, 31)); Np = int(rng.integers(40, 3000)); Nl = int(rng.integers(0, 600)); nsl = int(rng.integers(2, 7))
    tmax = int(rng.integers(3, min(K, 16) + 1)); tmin = int(rng.integers(2, tmax + 1))
    marg = bool(rng.integers(0, 2)); fused = [dict(), dict(lm_fused=0), dict(lm_fused=2)][int(rng.integers(0, 3))]
    rev = float(rng.choice([0.0, 0.0, 0.2]))
    seq = W.make_sequence(K, nsl + 1, Np, max(Nl, 1), seed=int(rng.integers(1, 1 << 30)), kf_dt=float(rng.choice([0.1, 0.25])), track=(tmin, tmax), revisit=rev)
    slid = pkg.new_problem(**fused)
    w_prev = res_prev = prior = None
    ok = True
    for i in range(nsl + 1):
        w = W.window_at(seq, i, K, prev=w_prev)
        if i == 0:
            wf = dict(w); slid.upload_window(w)
        else:
            wf = W.window_from_results(w, w_prev, res_prev)
            slid.slide_window(W.slide_delta(w_prev, w))
        for kind, d in w["huber"].items(): slid.set_robust(kind, True, d)
        slid.set_prior(prior)
        fresh = pkg.new_problem(**fused); wf["prior"] = prior; fresh.upload_window(wf)
        out = []
        for p in (slid, fresh):
            r = pkg.protocol.local_ba(p)
            pr = p.marginalize(0, pkg.protocol.MARG_NUM) if marg else None
            out.append((r["stage1"].chi2_final, r["stage2"].chi2_final, r["stage2"].trials, r["gated"], pkg.protocol.results(p), pr))
        fresh.close()
        a, b = out
        same = a[:4] == b[:4] and all(np.array_equal(a[4][k], b[4][k]) for k in a[4]) and (a[5] is None or all(np.array_equal(a[5][k], b[5][k]) for k in ("J0", "r0", "Ar", "br", "x0")))
        if not same:
            ok = False; print("MISMATCH case %d window %d: K %d Np %d Nl %d tracks %d..%d marg %s opts %s" % (case, i, K, Np, Nl, tmin, tmax, marg, fused), a[:4], b[:4], flush=True); break
        w_prev, res_prev, prior = w, a[4], a[5]
    slid.close()
    bad += not ok
    print("case %d: K %d Np %d Nl %d tracks %d..%d revisit %.1f marg %d opts %s slides %d: %s" % (case, K, Np, Nl, tmin, tmax, rev, marg, fused, nsl, "ok" if ok else "MISMATCH"), flush=True)
print("soak: %d cases, %d mismatches" % (N, bad))
sys.exit(1 if bad else 0)
