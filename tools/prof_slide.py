"""Steady-state BA calls on slid windows, step by step (configs[2] shape, or `realistic`); `--laps`: the library's own laps on stderr."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); W = pkg.window
real = "realistic" in sys.argv
K, Np, Nl, kw = (12, 2000, 400, dict(kf_dt=0.1, track=(6, 12), revisit=0.2)) if real else (50, 20000, 4000, {})
n = 6
seq = W.make_sequence(K, n + 1, Np, Nl, seed=0x5EED00E0 + K, **kw)
wins = [W.window_at(seq, 0, K)]
for i in range(1, n + 1): wins.append(W.window_at(seq, i, K, prev=wins[-1]))
deltas = [W.slide_delta(wins[i], wins[i + 1]) for i in range(n)]
opts = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}      # plba_options knobs, e.g. lm_fused=2
p = pkg.new_problem(diag=1 if "--laps" in sys.argv else 0, **opts)
p.upload_window(wins[0]); pkg.protocol.local_ba(p); pkg.protocol.results(p)
for i in range(n):
    w = wins[i + 1]; t = [time.perf_counter()]
    p.slide_window(deltas[i]); t.append(time.perf_counter())
    for kind, d in w["huber"].items(): p.set_robust(kind, True, d)
    t.append(time.perf_counter())
    s1 = p.optimize(5); t.append(time.perf_counter())
    p.gate_outliers(W.CHI2_GATE); t.append(time.perf_counter())
    s2 = p.optimize(10); t.append(time.perf_counter())
    kf = p.get_keyframes(); pts = p.get_points(); lns = p.get_lines(); t.append(time.perf_counter())
    d = np.diff(t) * 1e3
    print("slide %d: slide %.2f robust %.2f optimize(5) %.2f gate %.2f optimize(10) %.2f read-back %.2f  total %.2f ms" % (i, d[0], d[1], d[2], d[3], d[4], d[5], d.sum()), flush=True)
