"""per-phase device times (options.profile = 2: every phase bracketed by HIP events) of configs[N] for a list of option sets:
    python tools/phase_probe.py 3 - lm_group_steps=2 lm_group_steps=3      -> ms per iteration of [k_lm_schur launch, gather, dense solve, back_gemv.., trial]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
cfg = int(sys.argv[1])
variants = [{k: int(v) for k, v in (kv.split("=") for kv in a.split(",") if kv)} if a != "-" else {} for a in sys.argv[2:]] or [{}]
w = pkg.window.make_config(cfg)
names = ["linearize_launch", "factorisation", "schur/gather", "dense_solve", "backsub", "trial", "exchange", "lm_blocks"]
for v in variants:
    g = pkg.new_problem(profile=2, **v); g.upload_window(w)
    g.optimize(5); g.gate_outliers(); g.save_state()
    ph = np.zeros(8); n = 0
    for rep in range(6):
        g.restore_state(); s = g.optimize(10); ph += np.array(list(s.ms_phase)); n += s.trials
    print("%-28s" % (v or "(default)"), "  ".join("%s %.1f" % (a, 1e3 * b / n) for a, b in zip(names, ph)), "us per trial; groups", int(g.debug_get("lm_fused")[1]), flush=True)
    g.close()
