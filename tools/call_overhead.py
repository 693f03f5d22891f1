"""fixed cost of one plba_optimize call against its per-iteration cost: T(n) = a + b n, from replays of optimize(n) for several n
(the headline benchmark replays optimize(10), so a / 10 is part of every 'iteration' it reports)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
w = pkg.window.make_config(cfg)
g = pkg.new_problem(); g.upload_window(w)
g.optimize(5); g.gate_outliers(); g.save_state()
ns, ts = [2, 5, 10, 20, 40], []
for n in ns:
    best = 1e9
    for rep in range(12):
        g.restore_state()
        t0 = time.perf_counter(); s = g.optimize(n); dt = time.perf_counter() - t0
        if s.trials == n: best = min(best, dt)
    ts.append(best * 1e3)
    print("optimize(%2d): %.4f ms  (%.4f per iteration)" % (n, ts[-1], ts[-1] / n))
b, a = np.polyfit(ns, ts, 1)
print("fit: %.4f ms per call + %.4f ms per iteration" % (a, b))
t0 = time.perf_counter()
for _ in range(50): g.restore_state()
print("restore_state: %.4f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
