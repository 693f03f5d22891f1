#!/usr/bin/env python
"""bench.py — local-BA LM iterations/second on a synthetic EuRoC-shaped window (BASELINE.json metric).

A "step" is one outer Levenberg-Marquardt iteration of the local bundle adjustment (one g2o
OptimizationAlgorithmLevenberg::solve(): one linearisation + >= 1 damped trial solve) on a
device-resident window.  Protocol per window (reference call site src/mapHandler.cpp:6038-6069):
stage 1 optimize(5) with Huber + chi2/depth gating run once untimed; the timed region replays
stage 2 (optimize(10), no Huber on point/line edges) from the saved post-gating state until exactly
K iterations have run.

N = 1  : BASELINE configs[2] — 50 KF / 20k points / 4k lines + IMU (the metric's configuration); `--config 5` runs
         BASELINE configs[4] (200 KF / 200k points / 40k lines + IMU) on the one GPU instead, and the default run carries
         a short configs[4] leg as `config.configs4_single_gpu` (the N = 1 denominator of the scaling runs).
N > 1  : BASELINE configs[4], ONE problem whose landmarks are sharded over the N ranks, one all-reduce
         (RCCL over xGMI) of the structurally non-zero part of the reduced camera system per trial: strong scaling,
         `value` = GLOBAL LM iterations per second of that one window.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (spec), v_mfma_f64_16x16x4_f64


def stage1_and_gate(prob, pkg):
    prob.optimize(pkg.protocol.STAGE1_ITERS)
    prob.gate_outliers(pkg.window.CHI2_GATE)
    prob.save_state()


def csrc_sha16():
    """hash of the kernel sources: a PMC summary is only quoted for the build it was measured on"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "pl-inertial-slam_amd", "csrc", "*.h*"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_prefix, config5=False, split=False):
    """HBM-side bytes per launch of a kernel from the committed PMC passes (profiles/r05_pmc_traffic.json for configs[2],
    r05_config5_pmc_traffic.json for configs[4]: separate rocprofv3 --pmc runs of this same command — FETCH_SIZE, WRITE_SIZE and the L2 read requests by size —, gfx950
    correction applied, tools/rocpd_extract.py, tools/collect_profiles_r05.sh).  split: (read bytes, write bytes) instead of their sum.
    PMC counters cannot be collected from inside this process, so the file is tied to the build it came from by a hash of
    csrc/: None when the file is absent or was measured on different kernel sources (never a stale number)."""
    path = os.path.join(ROOT, "profiles", "r05_config5_pmc_traffic.json" if config5 else "r05_pmc_traffic.json")
    try:
        js = json.load(open(path))
        ks = js["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    if js.get("csrc_sha16") != csrc_sha16():
        return None
    for name, v in ks.items():
        if kernel_prefix in name and "traffic_bytes_per_launch" in v:
            if split:
                rd = v.get("fetch_bytes_exact")
                return (rd, v["traffic_bytes_per_launch"] - rd) if rd is not None else None
            return v["traffic_bytes_per_launch"]
    return None


def run_iterations(prob, n_iters, per_call=10):
    """Replay stage 2 from the saved state until exactly n_iters LM iterations ran."""
    done, trials, phases = 0, 0, np.zeros(8)
    guard = 0
    while done < n_iters:
        prob.restore_state()
        st = prob.optimize(min(per_call, n_iters - done))
        if st.iterations == 0:
            guard += 1
            if guard > 3:
                raise RuntimeError("optimize() makes no progress")
        done += st.iterations
        trials += st.trials
        phases += np.array(list(st.ms_phase))
    return done, trials, phases


def pose_delta(a, b, win):
    """max over the window keyframes of |dP|_inf (m), |dV|_inf (m/s), |Log(R_ref^T R)| (rad), |d(b+db)|_inf"""
    dphi = 0.0
    for qa, qb in zip(a["q"], b["q"]):
        dphi = max(dphi, float(np.linalg.norm(win.log_so3(win.R_from_quat(qb).T @ win.R_from_quat(qa)))))
    return dict(P_m=float(np.abs(a["P"] - b["P"]).max()), V_mps=float(np.abs(a["V"] - b["V"]).max()), rot_rad=dphi,
                bias=float(max(np.abs(a["dbg"] - b["dbg"]).max(), np.abs(a["dba"] - b["dba"]).max())))


def parity_and_config4(w, pkg):
    """The second half of BASELINE's metric (max |delta pose| vs the CPU path) on the benchmarked window: the whole
    reference protocol (5 + 10 iterations, gating) on the GPU and on the oracle; then configs[3]: marginalize the
    oldest keyframe on the GPU and time the next BA call, which carries the prior edge."""
    from oracle import oracle as orc
    g, o = pkg.new_problem(), orc.new_problem()
    g.upload_window(w); o.upload_window(w)
    pkg.protocol.local_ba(g); pkg.protocol.local_ba(o)
    d = pose_delta(g.get_keyframes(), o.get_keyframes(), pkg.window)
    d["landmarks_m"] = float(max(np.abs(g.get_points() - o.get_points()).max(), np.abs(g.get_lines() - o.get_lines()).max()))
    d["tolerance"] = 1e-5
    o.close()
    t_marg = None
    for _ in range(3):                  # best of three, as for the end-to-end call: the first one pays for module load and first-touch allocations
        t0 = time.perf_counter()
        prior = g.marginalize(0, pkg.protocol.MARG_NUM)
        dt_m = time.perf_counter() - t0
        t_marg = dt_m if t_marg is None else min(t_marg, dt_m)
    g.close()
    w2 = dict(w); w2["prior"] = prior
    g2 = pkg.new_problem()
    g2.upload_window(w2)
    g2.optimize(1)                      # structure build + first touch outside the timed call
    g2.upload_window(w2)
    t0 = time.perf_counter()
    r = pkg.protocol.local_ba(g2)
    t_ba = time.perf_counter() - t0
    its = r["stage1"].iterations + r["stage2"].iterations
    g2.close()
    c4 = dict(workload="configs[3]: configs[2] + marginalization prior", marginalize_ms=t_marg * 1e3, prior_dim=int(prior["n"]),
              ba_call_with_prior_ms=t_ba * 1e3, iterations=int(its), iterations_per_s=its / t_ba)
    return d, c4


def config5_leg(pkg, stream, iters=30):
    """BASELINE configs[4] (200 KF / 200k points / 40k lines + IMU) on ONE GPU: the N = 1 denominator of the N > 1 runs,
    which strong-scale exactly this window.  Same protocol as the main leg, fewer iterations."""
    import torch
    w = pkg.window.make_config(5)
    prob = pkg.new_problem()
    prob.set_stream(stream.cuda_stream)
    prob.upload_window(w)
    stage1_and_gate(prob, pkg)
    run_iterations(prob, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done, trials, _ = run_iterations(prob, iters)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = dict(workload="configs[4]: 200 KF / 200k points / 40k lines + IMU, one GPU", iterations_per_s=done / dt, ms_per_iteration=dt / done * 1e3,
               steps=done, trials_per_iteration=trials / max(done, 1), point_obs=int(w["meta"]["Ep"]), line_obs=int(w["meta"]["El"]),
               pose_dim=int(prob.debug_get("pose_dim")[0]), dense_dim=int(prob.debug_get("dense_dim")[0]))
    prob.close()
    return out


def _side_leg(pkg, stream, w, iters, **opts):
    """stage-2 LM iterations of one window, timed like the headline (replayed from the post-gating state), with the given options"""
    import torch
    prob = pkg.new_problem(**opts)
    prob.set_stream(stream.cuda_stream)
    prob.upload_window(w)
    stage1_and_gate(prob, pkg)
    run_iterations(prob, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done, trials, _ = run_iterations(prob, iters)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lmf = prob.debug_get("lm_fused")
    out = dict(iterations_per_s=done / dt, ms_per_iteration=dt / done * 1e3, trials_per_iteration=trials / max(done, 1),
               pose_dim=int(prob.debug_get("pose_dim")[0]), dense_dim=int(prob.debug_get("dense_dim")[0]),
               twin_factorisation=bool(prob.debug_get("twin")[0]), banded_twisted_solve=bool(prob.debug_get("band")[0]),
               fused_landmark_passes=bool(lmf[0]), wide_groups=int(lmf[3]) if len(lmf) > 3 else 0)
    prob.close()
    return out


def realistic_leg(pkg, stream, iters=100):
    """The reference's own window shape — 12 keyframes (include/mapHandler.h:217), tracks over 6 .. 12 of them (most of the window), one
    landmark in five seen again after a gap — instead of SURVEY 8d's generator (tracks over <= 8 consecutive keyframes).  Round 4: the
    fused landmark passes take such a window (wide groups for the tracks over 9 .. 12 keyframes); the leg reports the DEFAULT options'
    choice and rate, and both landmark paths forced, side by side."""
    w = pkg.window.make_window(12, 2000, 400, imu=True, seed=0x5EED00C0, kf_dt=0.1, track=(6, 12), revisit=0.2)
    out = dict(workload="12 KF / 2000 points / 400 lines + IMU, tracks over 6..12 keyframes, 20 % non-consecutive re-observations",
               point_obs=int(w["meta"]["Ep"]), line_obs=int(w["meta"]["El"]))
    out.update(_side_leg(pkg, stream, w, iters))
    out["record_based_passes"] = {k: v for k, v in _side_leg(pkg, stream, w, iters, lm_fused=0).items() if k in ("iterations_per_s", "ms_per_iteration")}
    f = _side_leg(pkg, stream, w, iters, lm_fused=2)
    out["fused_passes_forced"] = {k: f[k] for k in ("iterations_per_s", "ms_per_iteration", "fused_landmark_passes", "wide_groups")}
    return out


def long_track_leg(pkg, stream, iters=60):
    """the headline depends on the generator's band structure (tracks over <= 8 CONSECUTIVE keyframes give a reduced
    camera system with a band of two 32-column tiles).  The same 50 KF / 20k points / 4k lines + IMU with tracks over 2 .. 15 keyframes and
    one landmark in five seen again up to 30 keyframes later (a place revisited): a reduced system WITHOUT a usable band (the plain dense
    factorisation, one launch per 32 columns) and landmarks for standard and wide groups side by side.  Default options; the solver and
    landmark path are reported.  (With re-observation gaps of 1 .. 3 keyframes only, the band survives — three tiles — and the multi-chain
    factorisation still runs: 0.252 ms per iteration, measured in round 4.)"""
    w = pkg.window.make_window(50, 20000, 4000, imu=True, seed=0x5EED00D0, track=(2, 15), revisit=0.2, revisit_gap=(1, 30))
    out = dict(workload="50 KF / 20k points / 4k lines + IMU, tracks over 2..15 keyframes, 20 % re-observed after a gap of 1..30 keyframes (no band in the reduced system)",
               point_obs=int(w["meta"]["Ep"]), line_obs=int(w["meta"]["El"]))
    out.update(_side_leg(pkg, stream, w, iters))
    return out


def slide_leg(pkg, K, Np, Nl, n_slides=5, **wkw):
    """One reference-shaped BA call on a SLID window: the mapping thread's steady state (src/mapHandler.cpp:1178-1221: one BA per new
    keyframe on a window that differs from the previous one by one keyframe).  The previous window stays resident; plba_slide_window
    sends the new keyframe, its observations and the new landmarks (the caller's own bookkeeping — which observations are new — is
    prepared outside the timed region, as the map holds it); then robust kernels, 5 + 10 iterations with gating, write-back.
    Best of `n_slides` consecutive slides; next to it the same windows through fresh uploads."""
    import torch
    W = pkg.window
    seq = W.make_sequence(K, n_slides + 1, Np, Nl, seed=0x5EED00E0 + K, **wkw)
    wins = [W.window_at(seq, 0, K)]
    for i in range(1, n_slides + 1):
        wins.append(W.window_at(seq, i, K, prev=wins[-1]))
    deltas = [W.slide_delta(wins[i], wins[i + 1]) for i in range(n_slides)]
    p = pkg.new_problem()
    p.upload_window(wins[0])
    pkg.protocol.local_ba(p)
    res = pkg.protocol.results(p)
    best_slid, best_fresh, its = None, None, 0
    results = [res]
    for i in range(n_slides):      # the steady state: consecutive slides of one resident problem
        w = wins[i + 1]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p.slide_window(deltas[i])
        for kind, dlt in w["huber"].items():
            p.set_robust(kind, True, dlt)
        r = pkg.protocol.local_ba(p)
        results.append(pkg.protocol.results(p))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if i > 0:      # (the first slide builds the pose structure of the sequence's windows)
            best_slid = dt if best_slid is None else min(best_slid, dt)
        its = r["stage1"].iterations + r["stage2"].iterations
    for i in range(n_slides):      # the same windows, each through a fresh handle and a full upload
        wf = W.window_from_results(wins[i + 1], wins[i], results[i])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        q = pkg.new_problem()
        q.upload_window(wf)
        pkg.protocol.local_ba(q)
        pkg.protocol.results(q)
        torch.cuda.synchronize()
        dtf = time.perf_counter() - t0
        q.close()
        if i > 0:
            best_fresh = dtf if best_fresh is None else min(best_fresh, dtf)
    p.close()
    m = wins[-1]["meta"]
    return dict(workload="%d KF window slid by one keyframe (sequence windows: %d points / %d lines, %d + %d observations)" % (K, m["Np"], m["Nl"], m["Ep"], m["El"]),
                slid_window_ba_call_ms=best_slid * 1e3, fresh_upload_ba_call_ms=best_fresh * 1e3, iterations=int(its),
                added_point_obs=int(len(deltas[-1]["po_pt"])), added_points=int(len(deltas[-1]["points"])))


def facade_leg(w, reps=5):
    """One localBundleAdjustmentWithImuAndMarg-shaped call through BOUNDARY 1 — the reference's own call-site code (one `new` per
    vertex and edge, optimize(5), the gating loop over the edge objects, optimize(10), marginalization, write-back loops) compiled
    against include/g2o + include/plba_g2o, tools/localba_harness.cpp `time` — lap by lap, best of `reps` fresh optimizers.
    `facade_ba_call_ms` = optimize + gating + optimize + write-back (what `end_to_end_ba_call_ms` covers through the C ABI);
    graph construction and teardown (the call site's own object churn) are reported next to it."""
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import harness_io
    exe = harness_io.build_harness()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "window.bin")
        harness_io.write_window(w, path, do_marg=1, max_kf=len(w["kf"]["P"]))
        out = subprocess.run([exe, "time", path, str(reps)], check=True, capture_output=True, text=True, timeout=300).stdout
    return json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])


def cpu_baseline(w, pkg, budget_s=24.0):
    """The CPU restatement of the reference's g2o path timed on this box's host cores on a bounded sample of the same window
    (stage-2 LM iterations from the post-gating state), three ways:
      single_thread_faithful  oracle/plba_oracle.c as the parity tests use it (-O3 -march=x86-64-v3, no contraction, dense Cholesky):
                              one thread, as the reference's g2o build runs
      fast_single_thread      the same source built like the reference (-O3 -march=native, reference CMakeLists.txt:41) with the
                              reduced system factored inside its envelope (a sparse Cholesky, as g2o's LinearSolverEigen is)
      openmp                  ... and its edge / landmark loops spread over the host cores (OpenMP)
    `value` is the FASTEST of them (the OpenMP leg) with the threads it used: the >= 10x claim of north_star is made against that."""
    from oracle import oracle as orc
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)      # a 1-GPU box's CPU share is 16
    legs = (("single_thread_faithful", orc.new_problem, 1, 0.4), ("fast_single_thread", lambda: orc.new_fast_problem(threads=1), 1, 0.3),
            ("openmp", lambda: orc.new_fast_problem(threads=cores), cores, 0.3))
    out = {}
    for name, mk, thr, share in legs:
        p = mk()
        p.upload_window(w)
        stage1_and_gate(p, pkg)
        iters, t0 = 0, time.perf_counter()
        while True:
            p.restore_state()
            st = p.optimize(2)
            iters += st.iterations
            el = time.perf_counter() - t0
            if el > budget_s * share or iters >= 400:
                break
        p.close()
        out[name] = dict(value=iters / el, cores=thr, iterations=iters, seconds=el)
    best = out["openmp"]
    return dict(value=best["value"], unit="iterations/s", cores=best["cores"], kind="port",
                sample="%d stage-2 LM iterations of the same window on %d OpenMP threads (%.1f s); also %d on 1 thread, faithful build (%.1f s), %d on 1 thread, "
                       "native build with a sparse Cholesky (%.1f s); %d host cores present" % (
                    best["iterations"], best["cores"], best["seconds"], out["single_thread_faithful"]["iterations"], out["single_thread_faithful"]["seconds"],
                    out["fast_single_thread"]["iterations"], out["fast_single_thread"]["seconds"], os.cpu_count() or 0),
                single_thread_faithful=out["single_thread_faithful"]["value"], fast_single_thread=out["fast_single_thread"]["value"],
                openmp=dict(cores=best["cores"], value=best["value"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config5-leg", action="store_true", help="skip the short configs[4] single-GPU leg of the default run")
    ap.add_argument("--config", type=int, default=0, help="BASELINE config (1-based): 3 = configs[2] (default at N = 1), 5 = configs[4] (always at N > 1)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink landmark counts (debug only; invalidates the number)")
    ap.add_argument("--kf", type=int, default=0, help="override the keyframe count (debug only)")
    ap.add_argument("--fb", type=int, default=0, help="dense factorisation block width (32/64); 0 = library default")
    ap.add_argument("--no-chain", action="store_true", help="debug: chain_elim = 0 (dense path on the full 15-dims-per-keyframe system)")
    ap.add_argument("--wide", action="store_true", help="debug: wide_steps = 1 (64 columns per launch; measured slower)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # rehearsal on a one-GPU box only (never set by the driver): all ranks share cuda:0 and all-reduce through gloo
    rehearsal = os.environ.get("PLBA_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    ge.build_hip()
    pkg = ge.load_package()
    win = pkg.window

    # ---- workload -------------------------------------------------------------------------------
    cfg_idx = args.config if args.config else (3 if world == 1 else 5)
    if cfg_idx not in (3, 5) or (world > 1 and cfg_idx != 5):
        raise SystemExit("--config must be 3 (N = 1) or 5")
    cfg = dict(win.CONFIGS[cfg_idx])
    if cfg_idx == 3:
        name = "configs[2]: 50 KF / 20k points / 4k lines + IMU preintegration edges (9-DoF PVR + 6-DoF bias vertices)"
    else:
        name = "configs[4]: 200 KF / 200k points / 40k lines + IMU preintegration edges" + (
            ", landmarks sharded over %d ranks, one all-reduce of the reduced pose normal equations per LM trial" % world if world > 1 else ", one GPU")
    if args.kf:
        cfg["K"] = args.kf
    cfg["Np"] = max(1, int(cfg["Np"] * args.scale)); cfg["Nl"] = max(1, int(cfg["Nl"] * args.scale))
    w_full = win.make_window(cfg["K"], cfg["Np"], cfg["Nl"], imu=cfg["imu"], seed=0x5EED0000 + cfg_idx)
    w = win.shard_window(w_full, rank, world) if world > 1 else w_full

    stream = torch.cuda.Stream()
    # profile=1: two HIP events per LM trial (on the stream the kernels run on) bracket the dense factorisation launches,
    # the dominant kernel; the full per-phase table comes from a second, untimed problem below
    extra = dict(**({"factor_block": args.fb} if args.fb else {}), **({"chain_elim": 0} if args.no_chain else {}), **({"wide_steps": 1} if args.wide else {}))
    prob = pkg.new_problem(profile=1, **extra)
    prob.set_stream(stream.cuda_stream)
    prob.upload_window(w)
    # N > 1: the exchange hook is C++ on RCCL (include/plba_rccl.h: ncclAllReduce on the library's stream, a communicator of its
    # own, no Python inside the LM loop); torch.distributed only carries rank 0's ncclUniqueId.  Fallback (recorded in the
    # JSON line): the same collective through torch.distributed.all_reduce from a ctypes callback.
    xch, xch_kind = None, None
    if world > 1:
        if not rehearsal and os.environ.get("PLBA_BENCH_TORCH_EXCHANGE") != "1":
            try:
                ge.build_rccl()
                xch = pkg.distributed.RcclExchange(dist, rank, world)
                xch_kind = "libplba_rccl.so: ncclAllReduce (C++)"
            except Exception as e:      # noqa: BLE001 — any failure to bring the native hook up must not lose the run
                print("rank %d: native RCCL hook unavailable (%s); using torch.distributed" % (rank, e), file=sys.stderr)
                xch = None
        ok = torch.tensor([1 if xch is not None else 0], device="cpu" if rehearsal else "cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)          # all ranks or none
        if int(ok.item()) == 0:
            xch, xch_kind = None, "torch.distributed.all_reduce (%s) from a ctypes callback" % ("gloo, host-staged rehearsal" if rehearsal else "nccl = RCCL")

    def attach(pb):
        if world == 1:
            return
        if xch is not None:
            xch.attach(pb)
        else:
            pb.set_shard(rank, world, pkg.distributed.make_allreduce(dist, local_rank, stream, via_host=rehearsal))
    attach(prob)

    stage1_and_gate(prob, pkg)
    run_iterations(prob, max(args.warmup, 1))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    done, trials, phases = run_iterations(prob, args.steps)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device="cpu" if rehearsal else "cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # algorithmic bytes: 32 B per point observation, 40 B per line observation, landmarks 24 / 48 B
    mg = w_full["meta"]
    Ep, El = mg["Ep"], mg["El"]
    b_iter = 2 * (32 * Ep + 40 * El) + 3 * (24 * mg["Np"] + 48 * mg["Nl"])        # one LM iteration of the whole window (all ranks)
    bytes_lin = 32 * len(w["po_pt"]) + 40 * len(w["lo_ln"]) + 24 * len(w["points"]) + 48 * len(w["lines"])      # one k_linearize launch on THIS rank's shard
    P = int(prob.debug_get("pose_dim")[0])
    fb = args.fb if args.fb else 32
    Pdense = int(prob.debug_get("dense_dim")[0])                          # = P unless chain_elim reduced the dense part
    n_fact_launches = ((Pdense + 63) // 64) * 64 // fb + 1               # first-block launch + one per block step
    if Pdense < P and fb == 32:
        n_fact_launches -= 1          # chain elimination: the first diagonal tile is factored inside k_chain_schur, ahead of the events
    if fb == 32 and ((Pdense + 63) // 64) * 64 // 32 <= 32 and not args.wide:
        n_fact_launches -= 1          # explicit-inverse back-substitution: the last block step (panels only) is folded into k_back_gemv
    flops_fact = Pdense ** 3 / 3.0 + Pdense * Pdense                     # LL^T of the dense system + the forward solve riding along
    fact_ms = phases[1] / max(trials, 1) / n_fact_launches               # live: HIP events over the timed region
    # second, untimed pass with every phase bracketed (profile=2): phase table + the HBM-bound kernel's launch time
    prob2 = pkg.new_problem(profile=2, **extra)
    prob2.set_stream(stream.cuda_stream)
    prob2.upload_window(w)
    attach(prob2)
    stage1_and_gate(prob2, pkg)
    n_lin0 = prob2.debug_get("prof_lin_launches")[0]
    done2, trials2, phases2 = run_iterations(prob2, min(args.steps, 50))
    sync()
    n_lin = int(prob2.debug_get("prof_lin_launches")[0] - n_lin0)      # k_linearize<true> launches bracketed by the events behind phases2[0]
    prob2.close()
    lin_ms = phases2[0] / max(n_lin, 1)
    phase_names = ["linearize_launch", "factorisation_launches", "schur", "dense_solve", "backsub_update", "trial_errors", "exchange", "landmark_blocks_and_reductions"]
    per_iter = {k: float(v / max(done2, 1)) for k, v in zip(phase_names, phases2)}
    fused = bool(prob.debug_get("lm_fused")[0])      # the fused landmark-major passes (plba_lm_dev.h) run instead of the record-based ones
    hbm_kernel = "k_lm_schur<0" if fused else "k_linearize<true>"      # (k_lm_schur<0, false>: the instantiation without the wide groups' code)
    roof_hbm = dict(bound="hbm", kernel=("k_lm_schur<0> (fused landmark-major pass: residuals, Jacobians, Hll, damped inverse and the rank-k update of the pose blocks "
                                         "from the observation arrays; chain segments ride in the same launch)") if fused
                    else "k_linearize<true> (observation pass; IMU / prior edge blocks ride in the same launch)",
                    achieved=bytes_lin / (lin_ms * 1e-3) / 1e9 if lin_ms > 0 else None,
                    peak=HBM_PEAK_GBS, unit="GB/s", frac=(bytes_lin / (lin_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if lin_ms > 0 else None,
                    traffic=pmc_traffic(hbm_kernel, config5=(cfg_idx == 5)), algorithmic_bytes_per_launch=bytes_lin, avg_launch_ms=lin_ms, launches_timed=n_lin)
    rw = pmc_traffic(hbm_kernel, config5=(cfg_idx == 5), split=True)
    if rw:      # `algorithmic` counts the INPUTS; the writes are the pass's outputs (per-group Schur partials, chi2 per observation)
        roof_hbm["traffic_read_bytes"], roof_hbm["traffic_write_bytes"] = rw
    kname = "k_chol32" if fb == 32 else "k_chol_step"
    banded = bool(prob.debug_get("band")[0])
    twin = bool(prob.debug_get("twin")[0])
    if twin:
        # plba_dense.hip launch_twin_cholesky: the same block steps (k_chol32_list: both ends of the band per launch, then k_chol32 on the
        # middle), fewer of them; flops counted as the DENSE factorisation's would overstate what the banded system needs, so the banded count
        T = ((Pdense + 63) // 64) * 64 // 32
        n_fact_launches = int(prob.debug_get("fact_launches")[0])
        flops_fact = T * (1.0 / 3.0 + 3.0 + 12.0) * 32 ** 3
        fact_ms = phases[1] / max(trials, 1) / n_fact_launches
        kname = "k_chol32_list"

    if banded:
        # plba_band.hip: the two launches (k_band_fwd: factorisation + forward substitution from both ends, k_band_back) of the banded
        # twisted solve; banded LL^T with 3 sub-diagonal tiles: (1/3 + 3 + 12) * 32^3 flops per 32-column tile
        T = ((Pdense + 63) // 64) * 64 // 32
        n_fact_launches = 2
        flops_fact = T * (1.0 / 3.0 + 3.0 + 12.0) * 32 ** 3
        fact_ms = phases[1] / max(trials, 1) / n_fact_launches
        kname = "k_band_fwd"
    ach = flops_fact / n_fact_launches / (fact_ms * 1e-3) / 1e12 if fact_ms > 0 else None
    roof_mfma = dict(bound="mfma", kernel=("k_band_fwd + k_band_back: banded twisted fp64 LL^T of the reduced camera system from both ends in LDS (%d of the P=%d pose dims stay dense after the velocity/bias chain elimination, band of 3 tiles, %d launches per solve)" % (Pdense, P, n_fact_launches)) if banded else ("k_chol32_list / k_chol32: one block step of the two-ended fp64 LL^T of the banded reduced camera system (%d of the P=%d pose dims stay dense after the velocity/bias chain elimination; both ends of the band per launch, %d dependent launches per solve; flops = the banded factorisation's)" % (Pdense, P, n_fact_launches)) if twin else "%s: one block step of the dense fp64 LL^T of the reduced camera system (%d of the P=%d pose dims stay dense after the velocity/bias chain elimination, %d launches per solve)" % (kname, Pdense, P, n_fact_launches),
                     achieved=ach, peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=(ach / FP64_MFMA_PEAK_TFLOPS) if ach else None,
                     traffic=pmc_traffic(kname, config5=(cfg_idx == 5)), algorithmic_flops_per_launch=flops_fact / n_fact_launches, avg_launch_ms=fact_ms)
    # the factorisation launches are the largest single consumer of an iteration (profiles/r01_*_kernel_stats.csv)
    roofline = roof_mfma

    out = {
        "metric": "local-BA iterations/sec (50 KF, 20k pts, 4k lines, IMU)" if cfg_idx == 3 else "local-BA iterations/sec (200 KF, 200k pts, 40k lines, IMU)",
        "value": done / dt,
        "unit": "iterations/s",
        "n_gpus": world, "steps": done, "warmup": args.warmup, "ms_per_step": dt / done * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": name, "K": cfg["K"], "points": cfg["Np"], "lines": cfg["Nl"], "point_obs": int(Ep), "line_obs": int(El),
                   "pose_dim": P, "trials_per_iteration": trials / max(done, 1), "protocol": "stage-2 LM iterations (no Huber on point/line edges) replayed from the post-gating state",
                   "global_iterations_per_s": done / dt, "dense_dim": Pdense, "exchange": xch_kind, "banded_twisted_solve": banded, "twin_factorisation": twin, "fused_landmark_passes": fused,
                   "value_definition": "global LM iterations/s of ONE window (total work fixed as N grows: its landmarks are sharded over the N ranks)",
                   "algorithmic_bytes_per_iteration": b_iter,
                   "hbm_frac_whole_iteration": b_iter / (dt / done) / 1e9 / (HBM_PEAK_GBS * world)},
        "roofline": roofline,
        "roofline_hbm_kernel": roof_hbm,
        "phase_ms_per_iteration": per_iter,
        # what a scaling curve is made of (rank 0's profile = 2 pass, every phase bracketed by events): the landmark-parallel part shrinks
        # with N, the reduced-camera solve is replicated on every rank, the exchange is what N > 1 adds
        "scaling_phases_ms_per_iteration": {
            "sharded_landmark_side": per_iter["linearize_launch"] + per_iter["schur"] + per_iter["backsub_update"] + per_iter["trial_errors"] + per_iter["landmark_blocks_and_reductions"],
            "replicated_solve": per_iter["dense_solve"], "exchange": per_iter["exchange"]},
    }
    if world == 1:
        # end-to-end cost of one reference-shaped BA call incl. PCIe: SoA upload + structure build + 5+10 LM
        # iterations + gating + write-back (reported for DESIGN.md; never `value`).  The benchmarked problem stays alive
        # next to it (round 1 had to close it first: a ~20 ms stall, since traced to pageable host copies and fixed).
        e2e, r2 = None, None
        for _ in range(3):                  # best of three: the first call pays for first-touch allocations
            t1 = time.perf_counter()
            p2 = pkg.new_problem()
            p2.upload_window(w)
            r2 = pkg.protocol.local_ba(p2)
            pkg.protocol.results(p2)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            p2.close()
            e2e = dt2 if e2e is None else min(e2e, dt2)
        out["config"]["end_to_end_ba_call_ms"] = e2e * 1e3
        out["config"]["end_to_end_iterations_per_s"] = (r2["stage1"].iterations + r2["stage2"].iterations) / e2e
        if cfg_idx == 3:
            out["config"]["sliding_window_call"] = slide_leg(pkg, 50, 20000, 4000)
            out["config"]["sliding_window_call_12kf"] = slide_leg(pkg, 12, 2000, 400, kf_dt=0.1, track=(6, 12), revisit=0.2)
            # the mapping thread's steady state (one BA per new keyframe): the same call on a window slid by one keyframe (plba_slide_window)
            out["config"]["end_to_end_slid_window_ba_call_ms"] = out["config"]["sliding_window_call"]["slid_window_ba_call_ms"]
            out["config"]["end_to_end_slid_window_ba_call_ms_12kf"] = out["config"]["sliding_window_call_12kf"]["slid_window_ba_call_ms"]
        if rank == 0:
            try:
                out["config"].update(facade_leg(w_full))
                out["config"]["facade_over_c_abi"] = out["config"]["facade_ba_call_ms"] / out["config"]["end_to_end_ba_call_ms"]
            except Exception as e:      # noqa: BLE001 — a side leg must not lose the line
                out["config"]["facade_ba_call_ms"] = None
                out["config"]["facade_error"] = repr(e)[:300]
    if world == 1 and cfg_idx == 3 and not args.no_config5_leg:
        out["config"]["configs4_single_gpu"] = config5_leg(pkg, stream)
        out["config"]["realistic_12kf_window"] = realistic_leg(pkg, stream)
        out["config"]["long_track_50kf_window"] = long_track_leg(pkg, stream)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1 and cfg_idx == 3:
            out["max_pose_delta_vs_cpu"], out["config4_sliding_window"] = parity_and_config4(w_full, pkg)
            out["cpu_baseline"] = cpu_baseline(w_full, pkg)
        else:
            # N > 1: the CPU leg belongs to the N = 1 line (contract); configs[4]: one oracle iteration takes longer than the
            # whole bench budget (P = 2985 dense Cholesky + 1 M observations on one thread), its parity is tests/test_large_configs.py
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if prob is not None:
        prob.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
