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


class RcclExchange:
    """The exchange hook in C++ (include/plba_rccl.h, libplba_rccl.so): ncclAllReduce on the library's stream, bound to a
    communicator of its own.  torch.distributed is used ONCE, to hand rank 0's ncclUniqueId to the other ranks; after that
    no Python runs inside the LM loop (Problem.set_shard_native)."""

    def __init__(self, dist, rank, world, lib_path=None):
        """Bring-up in three steps, each followed by an all-ranks vote, so that a failure on ONE rank raises on ALL of them instead of
        leaving the others inside a collective nobody else enters: (1) local: dlopen, rank 0 makes the ncclUniqueId;
        (2) broadcast of the id; (3) ncclCommInitRank (itself a collective) and a vote on its result."""
        import os
        import torch
        self.comm = C.c_void_p()
        self.lib = None
        self.rank, self.world = rank, world
        on_gpu = world > 1 and dist.get_backend() == "nccl"
        dev = "cuda" if on_gpu else "cpu"

        def vote(ok, what, err):
            """MIN over the ranks of a success flag; every rank raises when any of them failed"""
            if world > 1:
                t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                all_ok = int(t.item()) == 1
            else:
                all_ok = ok
            if not all_ok:
                self.close()
                raise RuntimeError("%s failed on %s: %s" % (what, "this rank" if not ok else "another rank", err or "-"))

        err, idb = None, C.create_string_buffer(128)
        try:      # (1) local
            path = lib_path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libplba_rccl.so")
            self.lib = C.CDLL(path)
            self.lib.plba_rccl_last_error.restype = C.c_char_p
            self.lib.plba_rccl_unique_id.argtypes = [C.c_char_p]
            self.lib.plba_rccl_init.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_char_p]
            self.lib.plba_rccl_destroy.argtypes = [C.c_void_p]
            if rank == 0 and self.lib.plba_rccl_unique_id(idb) != 0:
                err = "plba_rccl_unique_id: %s" % self.lib.plba_rccl_last_error().decode()
        except OSError as e:
            err = "loading libplba_rccl.so: %s" % e
        vote(err is None, "RCCL exchange bring-up (library / unique id)", err)
        if world > 1:      # (2)
            t = torch.tensor(list(idb.raw), dtype=torch.uint8, device=dev)
            dist.broadcast(t, src=0)
            idb = C.create_string_buffer(bytes(t.cpu().tolist()), 128)
        rc = self.lib.plba_rccl_init(C.byref(self.comm), rank, world, idb)      # (3)
        vote(rc == 0, "plba_rccl_init", None if rc == 0 else self.lib.plba_rccl_last_error().decode())

    @property
    def fn_addr(self):
        return C.cast(self.lib.plba_rccl_allreduce, C.c_void_p).value

    def attach(self, problem):
        problem.set_shard_native(self.rank, self.world, self.fn_addr, self.comm.value)

    def allreduce(self, ptr, n, op, stream):
        """direct call of the native hook (tests)"""
        f = self.lib.plba_rccl_allreduce
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        if f(self.comm, C.c_void_p(ptr), n, op, C.c_void_p(stream)) != 0:
            raise RuntimeError("plba_rccl_allreduce: %s" % self.lib.plba_rccl_last_error().decode())

    def close(self):
        if self.comm and self.lib is not None:
            self.lib.plba_rccl_destroy(self.comm)
        self.comm = C.c_void_p()
