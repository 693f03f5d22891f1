"""A/B timing of plba_options knobs in ONE process (same box, same clocks; separate runs on a shared pool differ by +- 3 %):
    python tools/ab_opts.py 3 chain_seg=4 chain_seg=6 chain_seg=8        (config index, then one variant per argument: k=v[,k=v...]; "-" = defaults)
Each variant gets a fresh problem per round (the knobs are read in prepare()); rounds alternate; ms per LM trial of stage-2 iterations
replayed from the saved post-gating state, best of 8 replays per round; medians of five rounds."""
import sys, time
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
cfg = int(sys.argv[1])
variants = [{k: (float(v) if "." in v or "e" in v else int(v)) for k, v in (kv.split("=") for kv in a.split(",") if kv)} if a != "-" else {} for a in sys.argv[2:]]
w = pkg.window.make_config(cfg)
res = {i: [] for i in range(len(variants))}
for rnd in range(5):
    for i, v in enumerate(variants):
        g = pkg.new_problem(**v); g.upload_window(w)
        g.optimize(5); g.gate_outliers(); g.save_state()
        best = 1e9
        for rep in range(8):
            g.restore_state()
            t0 = time.perf_counter(); s = g.optimize(10); dt = time.perf_counter() - t0
            best = min(best, dt / max(s.trials, 1))
        res[i].append(best * 1e3)
        g.close()
for i, v in enumerate(variants):
    r = sorted(res[i])
    print("%-40s median %.4f  min %.4f  max %.4f ms per trial" % (v or "(default)", r[len(r) // 2], r[0], r[-1]))
