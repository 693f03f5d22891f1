"""One reference-shaped BA call at configs[2] (upload + 5 iterations + gating + 10 iterations + read-back), timed phase by phase;
`--laps` (options.diag bit 0) adds the laps of prepare() (host structure build) on stderr."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import __graft_entry__ as g

pkg = g.load_package()
# `realistic`: the reference's own window shape (12 keyframes, tracks over 6 .. 12 of them) instead of BASELINE configs[2]
w = pkg.window.make_window(12, 2000, 400, imu=True, seed=0x5EED00C0, kf_dt=0.1, track=(6, 12), revisit=0.2) if "realistic" in sys.argv else pkg.window.make_config(3)
for rep in range(int(os.environ.get("REPS", "4"))):
    t = [time.perf_counter()]
    p = pkg.new_problem(diag=1 if "--laps" in sys.argv else 0, **({"lm_fused": 2} if "--fused" in sys.argv else {"lm_fused": 0} if "--record" in sys.argv else {})); t.append(time.perf_counter())
    p.upload_window(w); t.append(time.perf_counter())
    s1 = p.optimize(5); t.append(time.perf_counter())
    p.gate_outliers(pkg.window.CHI2_GATE); t.append(time.perf_counter())
    s2 = p.optimize(10); t.append(time.perf_counter())
    kf = p.get_keyframes(); pts = p.get_points(); lns = p.get_lines(); t.append(time.perf_counter())
    p.close(); t.append(time.perf_counter())
    d = np.diff(t) * 1e3
    print("rep %d: create %.2f upload %.2f optimize(5) %.2f [device %.2f] gate %.2f optimize(10) %.2f [device %.2f] read-back %.2f close %.2f  total %.2f ms" %
          (rep, d[0], d[1], d[2], s1.ms_total, d[3], d[4], s2.ms_total, d[5], d[6], d.sum()), flush=True)
