"""Multi-GPU plumbing: the library asks the host for an in-place all-reduce of a device
buffer through the plba_allreduce_fn hook; here that hook is torch.distributed (backend "nccl" =
RCCL over xGMI on ROCm, "gloo" in the CPU tests).  Landmarks are sharded with window.shard_window;
no other collective exists on the path."""
import ctypes as C

import numpy as np


class _DevPtr:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 3}


def as_tensor(ptr, n, device_index):
    """View n doubles of device memory at `ptr` as a torch tensor (no copy)."""
    import torch
    return torch.as_tensor(_DevPtr(ptr, n), device=torch.device("cuda", device_index))


def host_array(ptr, n):
    """View n doubles of host memory at `ptr` as a numpy array (no copy)."""
    return np.ctypeslib.as_array(C.cast(C.c_void_p(ptr), C.POINTER(C.c_double)), shape=(int(n),))


def make_allreduce(dist, device_index=None, stream=None, via_host=False):
    """Build the callback for Problem.set_shard.  device_index=None: the buffer is host memory
    (gloo tests against a CPU implementation of plba.h); otherwise a device buffer all-reduced by
    RCCL, ordered on `stream` (the torch stream handed to plba_set_stream)."""
    import torch

    def _op(op):
        return dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX

    if device_index is None:
        def fn(ptr, n, op, _stream):
            t = torch.from_numpy(host_array(ptr, n))
            dist.all_reduce(t, op=_op(op))
        return fn

    if via_host:   # gloo over host staging: lets several ranks share one GPU in the tests
        def fn(ptr, n, op, _stream):
            t = as_tensor(ptr, n, device_index)
            if stream is not None:
                stream.synchronize()
            else:
                torch.cuda.synchronize()
            h = t.cpu()
            dist.all_reduce(h, op=_op(op))
            t.copy_(h)
            torch.cuda.synchronize()
        return fn

    def fn(ptr, n, op, _stream):
        t = as_tensor(ptr, n, device_index)
        if stream is not None:
            with torch.cuda.stream(stream):
                dist.all_reduce(t, op=_op(op))
        else:
            dist.all_reduce(t, op=_op(op))
    return fn
