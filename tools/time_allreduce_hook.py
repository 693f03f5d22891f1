"""Host-side cost of the all-reduce hook (torch.distributed on a device pointer) with a world-size-1 RCCL group."""
import os, sys, time
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
import __graft_entry__ as ge
pkg = ge.load_package()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
stream = torch.cuda.Stream()
fn = pkg.distributed.make_allreduce(dist, 0, stream)
buf = torch.zeros(300000, dtype=torch.float64, device="cuda")
small = torch.zeros(8, dtype=torch.float64, device="cuda")
for name, t, n in (("2.4 MB", buf, 300000), ("2 doubles", small, 2)):
    for _ in range(5): fn(t.data_ptr(), n, 0, stream.cuda_stream)
    stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): fn(t.data_ptr(), n, 0, stream.cuda_stream)
    t1 = time.perf_counter()
    stream.synchronize()
    t2 = time.perf_counter()
    print("%-10s host enqueue %.1f us per call, incl. device %.1f us per call" % (name, (t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
dist.destroy_process_group()
