"""Multi-GPU glue: one process per GPU, residual blocks sharded by landmark, one RCCL all-reduce (sum, fp64) of the reduced
normal equations per LM linearisation. torch.distributed is plumbing only: the library hands a device
pointer + count to the hook, the hook wraps it as a tensor and calls dist.all_reduce on the library's stream.
"""
from __future__ import annotations

import ctypes as C

from . import _lib


def _as_tensor(ptr: int, count: int, device):
    """Zero-copy fp64 view of library-owned device memory."""
    import torch

    class _Iface:
        __cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    return torch.as_tensor(_Iface(), device=device)


def attach_allreduce(problem, dist, group=None):
    """Registers the exchange hook on `problem` and agrees on the band layout across ranks. Returns an object that must be
    kept alive as long as the problem (it owns the ctypes callback)."""
    import torch
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    backend = dist.get_backend(group)
    on_gpu = backend == "nccl"
    device = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    # same band width everywhere
    problem._check(problem.lib.set_shard(problem.h, rank, world, 0), "set_shard")
    bw = torch.tensor([problem.lib.band_blocks(problem.h)], dtype=torch.int64, device=device)
    dist.all_reduce(bw, op=dist.ReduceOp.MAX, group=group)
    problem._check(problem.lib.set_shard(problem.h, rank, world, int(bw.item())), "set_shard")
    views = {}

    def hook(_user, ptr, count, stream):
        try:
            key = (ptr, count)
            t = views.get(key)
            if t is None:
                t = views[key] = _as_tensor(ptr, count, device)
            ext = torch.cuda.ExternalStream(stream) if stream else torch.cuda.current_stream()
            with torch.cuda.stream(ext):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            import sys
            print(f"hyperslam_amd all-reduce hook failed: {e!r}", file=sys.stderr)
            return 1

    cb = _lib.ALLREDUCE_FN(hook)
    problem._check(problem.lib.set_allreduce(problem.h, cb, None), "set_allreduce")
    problem._allreduce_cb = cb
    return cb
