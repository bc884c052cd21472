"""Multi-GPU glue: one process per GPU, residual blocks sharded by landmark, one RCCL all-reduce (sum, fp64) of the reduced
normal equations per LM linearisation. torch.distributed is plumbing only: the library hands a pointer +
count to the hook, the hook wraps it as a tensor (zero copy) and calls dist.all_reduce on the library's stream.

The same hook drives the oracle on CPU tensors with the gloo backend (tests/test_distributed_cpu.py).
"""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np

from . import _lib


def _device_tensor(ptr: int, count: int, device):
    """Zero-copy fp64 view of library-owned device memory."""
    import torch

    class _Iface:
        __cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    return torch.as_tensor(_Iface(), device=device)


def _host_tensor(ptr: int, count: int):
    import torch
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(count,))
    return torch.from_numpy(arr)


def attach_allreduce(problem, dist, group=None, device_memory=None):
    """Registers the exchange hook on `problem` and agrees on the band layout across ranks.

    device_memory: True if the library hands out GPU pointers (product library), False for host pointers (oracle);
    default: product library <=> True. With a backend that cannot reduce GPU tensors (gloo) the buffer is staged through
    the host. Returns the ctypes callback (kept alive on the problem)."""
    import torch
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    backend = dist.get_backend(group)
    if device_memory is None:
        device_memory = problem.lib.prefix == "hs_"
    device = torch.device("cuda", torch.cuda.current_device()) if device_memory else torch.device("cpu")
    comm_device = device if backend == "nccl" else torch.device("cpu")
    # same band width everywhere
    problem._check(problem.lib.set_shard(problem.h, rank, world, 0), "set_shard")
    # (max and min in one collective: a shard the library refuses — hs_band_blocks < 0, e.g. a track beyond the band limit that only this
    #  rank's landmarks reach — must stop EVERY rank here; a rank that raised on its own would leave the others waiting in the first exchange)
    local_bw = int(problem.lib.band_blocks(problem.h))
    bw = torch.tensor([local_bw, -local_bw], dtype=torch.int64, device=comm_device)
    dist.all_reduce(bw, op=dist.ReduceOp.MAX, group=group)
    if -int(bw[1].item()) < 0:
        raise RuntimeError("sharded window refused (invalid on at least one rank): " +
                           (problem.lib.last_error(problem.h).decode() if local_bw < 0 else "another rank's shard"))
    problem._check(problem.lib.set_shard(problem.h, rank, world, int(bw[0].item())), "set_shard")
    views = {}

    def hook(_user, ptr, count, stream):
        try:
            key = (ptr, count)
            t = views.get(key)
            if t is None:
                t = views[key] = _device_tensor(ptr, count, device) if device_memory else _host_tensor(ptr, count)
            if device_memory:
                ext = torch.cuda.ExternalStream(stream) if stream else torch.cuda.current_stream()
                with torch.cuda.stream(ext):
                    if backend == "nccl":
                        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                    else:  # host staging (tests on a single GPU with gloo)
                        h = t.cpu()
                        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                        t.copy_(h)
                        ext.synchronize()
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print(f"hyperslam_amd all-reduce hook failed: {e!r}", file=sys.stderr)
            return 1

    cb = _lib.ALLREDUCE_FN(hook)
    problem._check(problem.lib.set_allreduce(problem.h, cb, None), "set_allreduce")
    problem._allreduce_cb = cb
    return cb


def attach_rccl(problem, dist, group=None):
    """RCCL directly on the data path (no Python in the solve loop): torch.distributed is used once, to agree on the band layout and
    to hand rank 0's ncclUniqueId to the other ranks; the library then owns its communicator and enqueues ncclAllReduce on its own
    stream. Needs one GPU per rank. Returns True on every rank if every rank has a communicator, False on every rank otherwise
    (the caller then registers the generic hook, attach_allreduce)."""
    import torch
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    backend = dist.get_backend(group)
    comm_device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    problem._check(problem.lib.set_shard(problem.h, rank, world, 0), "set_shard")
    # (max and min in one collective: a shard the library refuses — hs_band_blocks < 0, e.g. a track beyond the band limit that only this
    #  rank's landmarks reach — must stop EVERY rank here; a rank that raised on its own would leave the others waiting in the first exchange)
    local_bw = int(problem.lib.band_blocks(problem.h))
    bw = torch.tensor([local_bw, -local_bw], dtype=torch.int64, device=comm_device)
    dist.all_reduce(bw, op=dist.ReduceOp.MAX, group=group)
    if -int(bw[1].item()) < 0:
        raise RuntimeError("sharded window refused (invalid on at least one rank): " +
                           (problem.lib.last_error(problem.h).decode() if local_bw < 0 else "another rank's shard"))
    problem._check(problem.lib.set_shard(problem.h, rank, world, int(bw[0].item())), "set_shard")
    # Every step is agreed on collectively so that a rank-local failure (librccl not loadable, communicator bootstrap refused)
    # cannot leave the other ranks waiting: the function returns the same boolean on every rank.
    buf = C.create_string_buffer(128)
    have_id = 1
    if rank == 0 and problem.lib.rccl_unique_id(buf) != 0:
        have_id = 0
    uid = torch.frombuffer(bytearray(buf.raw) + bytearray([have_id]), dtype=torch.uint8).to(comm_device)
    dist.broadcast(uid, src=0, group=group)
    raw = bytes(uid.cpu().numpy().tobytes())
    ok = 0
    if raw[128] == 1:
        ok = 1 if problem.lib.rccl_init(problem.h, raw[:128], rank, world) == 0 else 0
        if not ok:
            msg = problem.lib.last_error(problem.h)
            print(f"hyperslam_amd: hs_rccl_init failed on rank {rank}: {msg.decode() if msg else ''}", file=sys.stderr)
    flag = torch.tensor([ok], dtype=torch.int64, device=comm_device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    agreed = bool(flag.item())
    if not agreed:  # a rank that did get a communicator must not keep it: every rank has to issue the same collectives
        problem.lib.rccl_shutdown(problem.h)
    return agreed
