"""Multi-GPU host side: one process per GPU, torch.distributed (RCCL = backend
"nccl" on ROCm, over xGMI) for the one exchange step of the k-truss path.

What is sharded (BASELINE north_star "edge-range partitioned ...
RCCL all-reduce on the support vectors"): every rank holds the same graph; the
triangle-support phase is split by source-vertex range (each rank enumerates
the triangles of its own slice of the oriented CSR), and the per-edge partial
support vectors (|E|+1 int32) are summed over the ranks with ONE all-reduce.  Integer sums are order independent, so results are
bit-identical for any world size.  The incidence index, the peel and the
gather then run replicated: the peel is ~500 dependent sub-rounds of tens of
microseconds each, which a per-sub-round collective (>= ~20 us latency) cannot
speed up at this problem size -- see DESIGN.md section 6.

The C ABI takes the all-reduce as a plain C callback (komb_allreduce_fn); this
module supplies it.  With the "nccl" backend the reduction runs in place on the
device buffer; with "gloo" (CPU tests, or several ranks sharing one GPU) the
buffer is staged through host memory.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import ptr


def shard_range(n_tasks, rank, world):
    """Contiguous task range of `rank`; the same arithmetic as ktruss.hip."""
    return n_tasks * rank // world, n_tasks * (rank + 1) // world


def n_support_tasks(nv, verts_per_task=16):
    """Support-phase work items: blocks of 16 consecutive source vertices (kTriV)."""
    return (nv + verts_per_task - 1) // verts_per_task


class _RawDeviceI32:
    """Zero-copy view of a raw device pointer for torch.as_tensor.  The ABI's buffer is uint32[count]; it is viewed as
    int32 because torch reduces int32 on every backend, and a 32-bit two's-complement SUM is the same bits either way
    (the header says so: include/komb_accel.h, komb_allreduce_fn)."""

    def __init__(self, address, count):
        self.__cuda_array_interface__ = {
            "shape": (int(count),), "typestr": "<i4", "data": (int(address), False), "version": 3, "strides": None}


def allreduce_sum_(tensor, group=None):
    """In-place SUM all-reduce of an int32 tensor over the ranks (any backend)."""
    import torch.distributed as dist
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
    return tensor


def make_allreduce_callback(device_index, group=None):
    """ctypes callback implementing komb_allreduce_fn with torch.distributed."""
    import torch
    import torch.distributed as dist

    backend = dist.get_backend(group)

    def _cb(_user, dev_ptr, count):
        try:
            t = torch.as_tensor(_RawDeviceI32(dev_ptr, count), device=torch.device("cuda", device_index))
            if backend == "nccl":
                torch.cuda.synchronize(device_index)        # kernels queued by the library are done
                allreduce_sum_(t, group)
                torch.cuda.synchronize(device_index)
            else:                                           # gloo: stage through the host
                h = t.cpu()
                allreduce_sum_(h, group)
                t.copy_(h)
                torch.cuda.synchronize(device_index)
            return 0
        except Exception as exc:  # noqa: BLE001 - must not unwind through C
            import sys
            print(f"komb_amd.distributed: all-reduce failed: {exc!r}", file=sys.stderr)
            return 1

    return _lib.ALLREDUCE_FN(_cb)


def core_run_sharded(acc, group=None):
    """komb_core_run_sharded on this rank's KombAccel (live degrees owned by vertex range, the frontier exchanged every
    sub-round through the all-reduce callback); every rank must call it."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    cb = make_allreduce_callback(acc.device, group)
    acc._keepalive = cb
    acc._sync_env_options()
    rc = acc._lib.komb_core_run_sharded(acc._ctx, rank, world, ctypes.cast(cb, ctypes.c_void_p), None)
    acc._check(rc)


def truss_run_slice(acc, vmask=None, group=None):
    """komb_truss_run_slice with this process' rank / world size: every rank peels the whole graph, each materialises its
    slice of the canonical results; nothing is exchanged (gather_slices() below reassembles them where that is wanted)."""
    import torch.distributed as dist
    acc.truss_run_slice(dist.get_rank(group), dist.get_world_size(group), vmask)


def gather_slices(arr, group=None):
    """The whole result from the ranks' slices of komb_truss_run_slice: they are zero outside their slice, so a SUM
    all-reduce of the fetched host arrays is the concatenation (int32, in place; gloo or RCCL through torch)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(arr)
    if dist.get_backend(group) == "nccl":
        d = t.cuda(); dist.all_reduce(d, group=group); t.copy_(d.cpu())
    else:
        dist.all_reduce(t, group=group)
    return arr


def truss_run_sharded(acc, vmask=None, group=None, shard_peel=None):
    """komb_truss_run_sharded on this rank's KombAccel; every rank must call it.  shard_peel=True/False switches the
    sharded peel (komb_set_shard_peel) for this and the following runs; None leaves the context as it is."""
    import torch.distributed as dist
    if shard_peel is not None:
        acc.set_shard_peel(shard_peel)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if vmask is not None:
        vmask = np.ascontiguousarray(vmask, dtype=np.uint8)
    cb = make_allreduce_callback(acc.device, group)
    acc._keepalive = cb
    acc._sync_env_options()
    rc = acc._lib.komb_truss_run_sharded(acc._ctx, ptr(vmask), rank, world, ctypes.cast(cb, ctypes.c_void_p), None)
    acc._check(rc)
