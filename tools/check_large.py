"""Large-K sanity: K=200 keyframes (P ~ 3000, config-5 pose dimension) on one GPU vs the oracle on a reduced landmark set."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
from oracle import oracle as orc
pkg = ge.load_package()
t = time.time(); w = pkg.window.make_window(200, 6000, 1200, imu=True, seed=0x5EED0005); print("gen %.1fs" % (time.time() - t), w["meta"])
g = pkg.new_problem(profile=2); g.upload_window(w)
t = time.time(); sg = g.optimize(3); tg = time.time() - t
print("hip: %d iters %d trials %.1f ms/iter chi %.1f -> %.1f fails %d phases(ms/iter) %s" % (sg.iterations, sg.trials, tg / sg.iterations * 1e3, sg.chi2_initial, sg.chi2_final, sg.solver_failures, np.round(np.array(list(sg.ms_phase)) / sg.iterations, 3)))
t = time.time(); sg2 = g.optimize(3); tg = time.time() - t
print("hip(2nd call): %.2f ms/iter" % (tg / sg2.iterations * 1e3))
o = orc.new_problem(); o.upload_window(w)
t = time.time(); so = o.optimize(3); to = time.time() - t
print("oracle: %d iters %.1f ms/iter chi %.1f -> %.1f" % (so.iterations, to / so.iterations * 1e3, so.chi2_initial, so.chi2_final))
o.optimize(3)
kg, ko = g.get_keyframes(), o.get_keyframes()
print("max dP %.2e dV %.2e dq %.2e  dpts %.2e" % (np.abs(kg["P"] - ko["P"]).max(), np.abs(kg["V"] - ko["V"]).max(), np.abs(kg["q"] - ko["q"]).max(), np.abs(g.get_points() - o.get_points()).max()))
