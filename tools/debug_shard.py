"""debug helper: 2 ranks sharing cuda:0, single-iteration optimize calls, compare traces across ranks each step"""
import os, sys, socket
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def worker(rank, world, port, reps):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch, torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    w = pkg.window.make_window(8, 240, 50, imu=True, seed=0xD157)
    ws = pkg.window.shard_window(w, rank, world)
    for rep in range(reps):
        p = pkg.new_problem(); p.upload_window(ws)
        p.set_shard(rank, world, pkg.distributed.make_allreduce(dist, 0, None, via_host=True))
        for it in range(12):
            st = p.optimize(1)
            tr = p.trace()
            P = int(p.debug_get("pose_dim")[0])
            x = p.debug_get("x")[:P]
            kf = p.get_keyframes()
            mine = dict(tr=tr, xp=x.tobytes(), kf=kf["P"].tobytes() + kf["q"].tobytes(), chi=st.chi2_final, fail=st.solver_failures)
            allv = [None] * world
            dist.all_gather_object(allv, mine)
            a, b = allv
            same = (a["tr"] == b["tr"]) and a["xp"] == b["xp"] and a["kf"] == b["kf"]
            if not same:
                if rank == 0:
                    xa, xb = np.frombuffer(a["xp"]), np.frombuffer(b["xp"])
                    print("REP", rep, "ITER", it, "MISMATCH tr_same", a["tr"] == b["tr"], "x maxdiff", np.abs(xa - xb).max(), "at", int(np.abs(xa - xb).argmax()), "of", P, "kf_same", a["kf"] == b["kf"])
                    print(" r0", a["tr"]); print(" r1", b["tr"]); sys.stdout.flush()
                break
        else:
            if rank == 0: print("rep", rep, "ok", st.chi2_final); sys.stdout.flush()
        p.close()
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port, int(sys.argv[1]) if len(sys.argv) > 1 else 6), nprocs=2, join=True)
