"""How long is the gap between two DEPENDENT kernels on this device — launched one by one on a stream WITH THE HOST AHEAD (a long kernel
in front keeps the queue full), and replayed from a hipGraph?  (torch is only the launcher here: 1-element adds.)  Decides whether
capturing an LM iteration in a graph could pay."""
import torch
x = torch.zeros(1, device="cuda")
big = torch.randn(8192, 8192, device="cuda")
s = torch.cuda.Stream()
N = 1000
with torch.cuda.stream(s):
    for _ in range(200): x.add_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(3):
        y = big @ big; y = big @ big; y = big @ big      # ~ tens of ms: the host enqueues everything below meanwhile
        e0.record(s)
        for _ in range(N): x.add_(1.0)
        e1.record(s)
        torch.cuda.synchronize()
        print("stream launches, queue full : %.2f us per dependent kernel" % (e0.elapsed_time(e1) * 1e3 / N))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(N): x.add_(1.0)
    g.replay(); torch.cuda.synchronize()
    for rep in range(3):
        e0.record(s); g.replay(); e1.record(s); torch.cuda.synchronize()
        print("graph replay                : %.2f us per dependent kernel node" % (e0.elapsed_time(e1) * 1e3 / N))
