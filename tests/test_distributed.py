"""N > 1: landmarks sharded over ranks, pose-side edges on rank 0, one all-reduce of the reduced
camera system per LM trial.  World size 2 over gloo:
  * CPU (not gpu): the oracle implementation of plba.h on each rank's shard == the unsharded oracle.
  * GPU: two processes share cuda:0, HIP problems on the shards, gloo through host staging."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


WIN_SMALL = (8, 240, 50, 0xD157, 5, 10)          # K, points, lines, seed, stage-1 / stage-2 iterations
WIN_K200 = (200, 6000, 1200, 0x5EED0005, 2, 2)    # the BASELINE configs[4] window shape (P = 2985) at a landmark count the oracle finishes
WIN_PRIOR = (12, 360, 80, 0xD158, 5, 10, True)      # ... with the marginalization prior of a previous slide (rank 0 owns it; forced separators in the chain)
WIN_K200_PRIOR = (200, 6000, 1200, 0x5EED0006, 2, 2, True)      # configs[4] in its stated form: 200 keyframes + prior, sharded
WIN_LONG = (12, 600, 120, 0x5EED00C0, 5, 10, False, dict(kf_dt=0.1, track=(6, 12), revisit=0.2))      # the reference's window shape: tracks over 9 .. 12 keyframes take the wide groups (round 4)


def _window(pkg, orc, win):
    """the (unsharded) window of a case; with a prior: the one the ORACLE's marginalization of the same window's first BA leaves —
    deterministic, so every rank and the reference derive the same"""
    kw = win[7] if len(win) > 7 else {}
    w = pkg.window.make_window(win[0], win[1], win[2], imu=True, seed=win[3], **kw)
    if len(win) > 6 and win[6]:
        o = orc.new_problem(); o.upload_window(w); pkg.protocol.local_ba(o, stage1=2, stage2=2); pr = o.marginalize(0, 50); o.close()
        w = pkg.window.make_window(win[0], win[1], win[2], imu=True, seed=win[3], **kw); w["prior"] = pr
    return w


def _worker(rank, world, port, use_hip, out, win=WIN_SMALL, opts=None):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    from oracle import oracle as orc
    pkg = ge.load_package()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = _window(pkg, orc, win)
        ws = pkg.window.shard_window(w, rank, world)
        if use_hip:
            torch.cuda.set_device(0)
            p = pkg.new_problem(**(opts or {}))
            p.upload_window(ws)
            p.set_shard(rank, world, pkg.distributed.make_allreduce(dist, 0, None, via_host=True))
        else:
            p = orc.new_problem()
            p.upload_window(ws)
            p.set_shard(rank, world, pkg.distributed.make_allreduce(dist))
        r = pkg.protocol.local_ba(p, stage1=win[4], stage2=win[5])
        res = pkg.protocol.results(p)
        tr = p.trace()
        pidx = ws["shard"]["pt_index"]
        fused = int(p.debug_get("lm_fused")[0]) if use_hip else 0
        out.put((rank, res["P"], res["V"], res["q"], res["dbg"], res["points"], pidx, r["gated"], r["stage2"].chi2_final,
                 [t["accepted"] for t in tr], fused))
        p.close()
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as e:   # report instead of dead-locking the peer in a collective
        import traceback
        out.put(("error", rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
        out.close(); out.join_thread()
        os._exit(1)


def _run(world, use_hip, win=WIN_SMALL, opts=None):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, use_hip, q, win, opts)) for r in range(world)]
    for p in procs:
        p.start()
    got = []
    try:
        for _ in range(world):
            g = q.get(timeout=240)
            if g[0] == "error":
                raise AssertionError("rank %d failed:\n%s" % (g[1], g[2]))
            got.append(g)
    finally:
        for p in procs:
            p.join(20)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    return sorted(got, key=lambda g: g[0])


def _check(got, pkg, orc, win=WIN_SMALL):
    w = _window(pkg, orc, win)
    ref = orc.new_problem(); ref.upload_window(w)
    rr = pkg.protocol.local_ba(ref, stage1=win[4], stage2=win[5])
    res = pkg.protocol.results(ref)
    # every rank holds the same keyframe estimates, equal to the unsharded solve
    for g in got:
        assert np.abs(g[1] - res["P"]).max() < 1e-8 and np.abs(g[2] - res["V"]).max() < 1e-8
        assert np.abs(g[3] - res["q"]).max() < 1e-8 and np.abs(g[4] - res["dbg"]).max() < 1e-8
        assert np.abs(g[5] - res["points"][g[6]]).max() < 1e-7      # (the shard's points: window.shard_window, time-contiguous stretches)
        assert g[8] == pytest.approx(rr["stage2"].chi2_final, rel=1e-8)
    assert got[0][1].tobytes() == got[1][1].tobytes()            # bit-identical pose blocks across ranks
    assert got[0][9] == got[1][9]                                # identical LM decisions
    assert tuple(sum(g[7][i] for g in got) for i in range(2)) == rr["gated"]
    ref.close()


def test_sharded_oracle_world2_gloo(pkg, orc):
    _check(_run(2, False), pkg, orc)


def test_sharded_oracle_world2_gloo_with_prior(pkg, orc):
    """the oracle's own sharded mode with a marginalization prior (owned by rank 0) == its unsharded solve"""
    _check(_run(2, False, WIN_PRIOR), pkg, orc, WIN_PRIOR)


@pytest.mark.gpu
def test_sharded_hip_world2_gloo_host_staged(pkg, orc, hip):
    _check(_run(2, True), pkg, orc)


@pytest.mark.gpu
@pytest.mark.parametrize("win", [WIN_SMALL, WIN_K200, WIN_LONG])
def test_sharded_hip_world2_fused_landmark_passes(pkg, orc, hip, win):
    """the fused landmark-major passes on two landmark shards (lm_fused = 2: these windows are below the default's 40 k observations):
    groups of the local landmarks only, gather + structural all-reduce, chain elimination behind the exchange, the trial launch
    without a riding decision, [chi2, scale] all-reduced — equal to the unsharded oracle, bit-identical across the ranks"""
    got = _run(2, True, win, {"lm_fused": 2})
    assert all(g[10] == 1 for g in got), "the fused passes were meant to run"
    _check(got, pkg, orc, win)


@pytest.mark.gpu
@pytest.mark.parametrize("win,opts", [(WIN_PRIOR, None), (WIN_PRIOR, {"lm_fused": 2}), (WIN_K200_PRIOR, {"lm_fused": 2})])
def test_sharded_hip_world2_with_marginalization_prior(pkg, orc, hip, win, opts):
    """BASELINE configs[4] in its stated form is a SHARDED window WITH the prior of the previous slide: rank 0 owns the prior edge and its
    constant Hessian block, the chain elimination keeps the prior's keyframes as separators, the structural exchange list carries the
    prior block — record-based passes and fused passes, small window and the 200-keyframe one, against the unsharded oracle"""
    got = _run(2, True, win, opts)
    if opts: assert all(g[10] == 1 for g in got)
    _check(got, pkg, orc, win)


@pytest.mark.gpu
def test_sharded_hip_world2_k200(pkg, orc, hip):
    """the configs[4] window shape (200 keyframes, P = 2985: structural exchange list of ~0.8 M doubles, chain elimination
    after the all-reduce) on two landmark shards == the unsharded oracle"""
    _check(_run(2, True, WIN_K200), pkg, orc, WIN_K200)


def test_shard_window_partitions_landmarks(pkg):
    w = pkg.window.make_window(6, 101, 23, imu=True, seed=3)
    parts = [pkg.window.shard_window(w, r, 3) for r in range(3)]
    assert sum(len(p["points"]) for p in parts) == 101 and sum(len(p["lines"]) for p in parts) == 23
    assert sum(len(p["po_pt"]) for p in parts) == len(w["po_pt"])
    for p in parts:
        assert p["imu"] is w["imu"] and (np.diff(p["po_pt"]) >= 0).all()
        assert p["po_pt"].max() < len(p["shard"]["pt_index"])
    allp = np.concatenate([p["shard"]["pt_index"] for p in parts])
    assert np.array_equal(np.sort(allp), np.arange(101))
    # time-contiguous: a later rank's points start no earlier than the previous rank's
    first = pkg.window._first_kf(w["po_pt"].astype(np.int64), w["po_kf"].astype(np.int64), 101)
    assert first[parts[0]["shard"]["pt_index"]].max() <= first[parts[1]["shard"]["pt_index"]].min() <= first[parts[1]["shard"]["pt_index"]].max() <= first[parts[2]["shard"]["pt_index"]].min()
    blocks = [pkg.window.shard_window(w, r, 3, by="index") for r in range(3)]
    assert np.array_equal(np.concatenate([b["shard"]["pt_index"] for b in blocks]), np.arange(101))


def _nccl_worker(port, out):
    """world-size-1 RCCL group on cuda:0: the exact callback bench.py hands to plba_set_shard (device pointer viewed as
    a torch tensor, all-reduce ordered on the library's stream) — the part of the N > 1 path gloo cannot cover."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        import torch
        import torch.distributed as dist
        import __graft_entry__ as ge
        pkg = ge.load_package()
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        stream = torch.cuda.Stream()
        probe = torch.ones(8, device="cuda")
        dist.all_reduce(probe)                                   # RCCL bring-up itself (communicator, first collective)
        torch.cuda.synchronize()
        out.put(("stage", "rccl_ready"))
        fn = pkg.distributed.make_allreduce(dist, 0, stream)
        with torch.cuda.stream(stream):
            buf = torch.arange(4096, dtype=torch.float64, device="cuda") * 0.5
            buf2 = buf.clone()
        fn(buf.data_ptr(), 4096, 0, stream.cuda_stream)          # sum over one rank: identity, in place
        fn(buf2.data_ptr() + 8 * 16, 64, 1, stream.cuda_stream)  # max on an interior slice (pointer arithmetic as the library does)
        stream.synchronize()
        ok = bool(torch.equal(buf.cpu(), torch.arange(4096, dtype=torch.float64) * 0.5) and torch.equal(buf2, buf))
        # a whole sharded solve through the same hook (world 1: the exchange calls are skipped by the library, the
        # set_shard / set_stream plumbing is not)
        w = pkg.window.make_window(6, 150, 30, imu=True, seed=0xD158)
        p = pkg.new_problem()
        p.set_stream(stream.cuda_stream)
        p.upload_window(pkg.window.shard_window(w, 0, 1))
        p.set_shard(0, 1, fn)
        st = p.optimize(3)
        p.close()
        dist.destroy_process_group()
        out.put(("ok", ok, st.iterations))
    except BaseException as e:
        import traceback
        out.put(("error", "".join(traceback.format_exception(type(e), e, e.__traceback__))))
        out.close(); out.join_thread()
        os._exit(1)


@pytest.mark.gpu
def test_rccl_allreduce_hook_on_device_pointers(pkg, hip):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    pr.start()
    import queue
    try:
        try:
            g = q.get(timeout=150)
        except queue.Empty:
            # the worker never got a plain torch all-reduce through: RCCL did not come up on this box (seen once in ~10
            # runs on the shared pool); nothing of this repository has run yet at that point
            pytest.skip("RCCL bring-up (init_process_group / first all_reduce) did not finish within 150 s on this box")
        if g[0] == "stage":
            g = q.get(timeout=240)
    finally:
        pr.join(timeout=60)
        if pr.is_alive():
            pr.kill()
    assert g[0] == "ok", g[-1]
    assert g[1] is True and g[2] >= 1


def _native_rccl_worker(out):
    """libplba_rccl.so on a world-size-1 communicator of its own (no torch.distributed at all): unique id, ncclCommInitRank,
    the hook called the way plba_api.hip calls it (device pointer, element count, op, stream), and a problem that carries it."""
    sys.path.insert(0, ROOT)
    try:
        import torch
        import __graft_entry__ as ge
        pkg = ge.load_package()
        ge.build_rccl()
        torch.cuda.set_device(0)
        x = pkg.distributed.RcclExchange(None, 0, 1)
        out.put(("stage", "rccl_ready"))
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            buf = torch.arange(1 << 16, dtype=torch.float64, device="cuda") * 0.25
            ref = buf.clone()
        x.allreduce(buf.data_ptr(), buf.numel(), 0, stream.cuda_stream)            # sum over one rank: identity
        x.allreduce(buf.data_ptr() + 8 * 100, 1000, 1, stream.cuda_stream)         # max on an interior slice
        stream.synchronize()
        ok = bool(torch.equal(buf, ref))
        w = pkg.window.make_window(6, 150, 30, imu=True, seed=0xD158)
        p = pkg.new_problem()
        p.set_stream(stream.cuda_stream)
        p.upload_window(pkg.window.shard_window(w, 0, 1))
        x.attach(p)
        st = p.optimize(3)
        p.close()
        x.close()
        out.put(("ok", ok, st.iterations))
    except BaseException as e:
        import traceback
        out.put(("error", "".join(traceback.format_exception(type(e), e, e.__traceback__))))
        out.close(); out.join_thread()
        os._exit(1)


@pytest.mark.gpu
def test_native_rccl_hook(pkg, hip):
    import queue
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_native_rccl_worker, args=(q,))
    pr.start()
    try:
        try:
            g = q.get(timeout=150)
        except queue.Empty:
            pytest.skip("RCCL bring-up (ncclCommInitRank) did not finish within 150 s on this box")
        if g[0] == "stage":
            g = q.get(timeout=240)
    finally:
        pr.join(timeout=60)
        if pr.is_alive():
            pr.kill()
    assert g[0] == "ok", g[-1]
    assert g[1] is True and g[2] >= 1


def _bringup_vote_worker(rank, world, port, out):
    """rank 1 cannot load the exchange library: BOTH ranks must raise, after the same number of collectives (no rank left waiting)"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        import torch.distributed as dist
        import __graft_entry__ as ge
        pkg = ge.load_package()
        dist.init_process_group("gloo", rank=rank, world_size=world)
        raised = None
        try:
            pkg.distributed.RcclExchange(dist, rank, world, lib_path="/nonexistent/libplba_rccl.so" if rank == 1 else os.path.join(ROOT, "tests", "conftest.py"))
        except RuntimeError as e:      # (rank 0's "library" is not loadable either — a text file: OSError -> vote; what matters is that both raise)
            raised = str(e)
        dist.barrier()      # both ranks get here: nobody is stuck in a broadcast the other never entered
        out.put((rank, raised))
        dist.destroy_process_group()
    except BaseException as e:
        import traceback
        out.put(("error", rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
        out.close(); out.join_thread()
        os._exit(1)


def test_rccl_bringup_failure_on_one_rank_raises_on_all(pkg):
    """a bring-up failure on one rank used to leave the others in dist.broadcast.  Now every step is followed by an
    all-ranks vote (distributed.RcclExchange.__init__); world size 2 over gloo, no GPU needed."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bringup_vote_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = []
    try:
        for _ in range(2):
            g = q.get(timeout=120)
            assert g[0] != "error", g[-1]
            got.append(g)
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert all(r[1] is not None and "failed" in r[1] for r in got), got
