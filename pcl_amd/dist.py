"""Multi-GPU host side of the ICP hot path: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI; "gloo" for the CPU tests).

The path shards by SOURCE points: correspondences are independent per source point, every rank
searches its own contiguous slab of the source against the full target index (replicated -- it fits
one GPU's 288 GB many times over; north_star shards the target only when it does not), and the only
exchange per iteration is the sum of the 32-double reduction record (27 normal-system terms, sum of
squared distances, counts): 256 bytes, latency-bound, one all-reduce.  Every rank then solves the
same 6x6 system and applies the same transform, so the ranks stay in lock step without a broadcast.
"""
import numpy as np

from ._lib import NSUMS


def shard_range(n_total, rank, world):
    """Contiguous slab [start, start+count) of `n_total` items for `rank`; slabs are disjoint, cover
    everything, and differ in size by at most one."""
    base, rem = divmod(int(n_total), int(world))
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def reduce_record_host(record, group=None):
    """All-reduce (sum) a host copy of the reduction record -- gloo path / tests."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(record, np.float64).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.numpy()


def device_doubles(ptr, count, device_index):
    """torch tensor aliasing `count` doubles at device pointer `ptr` (no copy)."""
    import torch

    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                  "version": 3}
    return torch.as_tensor(h, device="cuda:%d" % device_index)


def make_allreduce_hook(device_index, group=None):
    """Hook for IterativeClosestPoint.setAllReduce: sums the DEVICE record in place over RCCL.  The
    context must have been created on torch's current stream so the collective is ordered after
    the kernel that produced the record and before the host read."""
    import torch.distributed as dist

    def hook(ptr, count, stream):
        assert count == NSUMS
        dist.all_reduce(device_doubles(ptr, count, device_index), op=dist.ReduceOp.SUM, group=group)
        return 0
    return hook
