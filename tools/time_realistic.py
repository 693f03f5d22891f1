"""ms per LM iteration of the reference-shaped 12-keyframe window (bench.py's realistic leg) and of small BASELINE-like windows with the
record-based passes (lm_fused = 0) and the fused ones (lm_fused = 2): where does the default threshold (options.lm_fused_min_obs) belong?"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
W = pkg.window
cases = [("12 KF 2000/400 tracks 6..12 revisit", lambda: W.make_window(12, 2000, 400, imu=True, seed=0x5EED00C0, kf_dt=0.1, track=(6, 12), revisit=0.2)),
         ("12 KF 600/120 tracks 6..12", lambda: W.make_window(12, 600, 120, imu=True, seed=0x5EED00C0, kf_dt=0.1, track=(6, 12), revisit=0.2)),
         ("12 KF 300/60 tracks 2..8", lambda: W.make_window(12, 300, 60, imu=True, seed=0x5EED00AA)),
         ("configs[0] 10 KF 2000/500 no IMU", lambda: W.make_config(1)),
         ("20 KF 4000/800", lambda: W.make_window(20, 4000, 800, imu=True, seed=3)),
         ("configs[2] x 0.2", lambda: W.make_config(3, scale=0.2))]
for name, mk in cases:
    w = mk()
    res = {}
    for rnd in range(3):
        for lmf in (0, 2):
            g = pkg.new_problem(lm_fused=lmf); g.upload_window(w)
            g.optimize(5); g.gate_outliers(); g.save_state()
            best = 1e9
            for rep in range(8):
                g.restore_state()
                t0 = time.perf_counter(); s = g.optimize(10); dt = time.perf_counter() - t0
                best = min(best, dt / max(s.trials, 1))
            res.setdefault(lmf, []).append(best * 1e3)
            fused = g.debug_get("lm_fused")
            g.close()
    print("%-40s obs %6d  record %.4f  fused %.4f ms/trial  (fused ran: %d, wide groups %d)" % (name, w["meta"]["Ep"] + w["meta"]["El"], sorted(res[0])[1], sorted(res[2])[1], fused[0], fused[3]), flush=True)
