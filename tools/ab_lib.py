"""A/B of two builds of libplba_hip.so inside ONE process (separate gpurun boxes differ by +-20 % in host speed): steady-state BA calls
on slid windows and on fresh uploads, alternating the two libraries, medians per step.
  python tools/ab_lib.py pl-inertial-slam_amd/libplba_base.so pl-inertial-slam_amd/libplba_hip.so [realistic]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); W = pkg.window
libs = [pkg.abi.Lib(os.path.abspath(a), "plba_") for a in sys.argv[1:3]]
real = "realistic" in sys.argv
K, Np, Nl, kw = (12, 2000, 400, dict(kf_dt=0.1, track=(6, 12), revisit=0.2)) if real else (50, 20000, 4000, {})
n = 8
seq = W.make_sequence(K, n + 1, Np, Nl, seed=0x5EED00E0 + K, **kw)
wins = [W.window_at(seq, 0, K)]
for i in range(1, n + 1): wins.append(W.window_at(seq, i, K, prev=wins[-1]))
deltas = [W.slide_delta(wins[i], wins[i + 1]) for i in range(n)]
probs = [pkg.abi.Problem(l, diag=1 if "--laps" in sys.argv else 0) for l in libs]
has_slide = ["slide_window" in l.fn for l in libs]
for p in probs:
    p.upload_window(wins[0]); pkg.protocol.local_ba(p); pkg.protocol.results(p)
rows = [[], []]
for i in range(n):
    w = wins[i + 1]
    for j, p in enumerate(probs):
        if "--laps" in sys.argv: print("---- lib %d window %d" % (j, i), file=sys.stderr, flush=True)
        t = [time.perf_counter()]
        if has_slide[j]: p.slide_window(deltas[i])
        else: p.upload_window(w)
        t.append(time.perf_counter())
        for kind, d in w["huber"].items(): p.set_robust(kind, True, d)
        s1 = p.optimize(5); t.append(time.perf_counter())
        p.gate_outliers(W.CHI2_GATE); t.append(time.perf_counter())
        s2 = p.optimize(10); t.append(time.perf_counter())
        p.get_keyframes(); p.get_points(); p.get_lines(); t.append(time.perf_counter())
        if i >= 2: rows[j].append(np.diff(t) * 1e3)
for j in range(2):
    m = np.median(np.array(rows[j]), axis=0)
    print("%-45s %s %.2f | optimize(5) %.2f | gate %.2f | optimize(10) %.2f | read-back %.2f | total %.2f ms" % (
        os.path.basename(sys.argv[1 + j]), "slide" if has_slide[j] else "upload", m[0], m[1], m[2], m[3], m[4], m.sum()))
