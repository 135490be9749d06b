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
    """Hook for IterativeClosestPoint.setAllReduce: sums the DEVICE record in place over RCCL, ON THE
    STREAM THE C SIDE PASSES (the context's stream, where the record was just produced and from which the
    host copy is issued afterwards).  The collective is enqueued under torch.cuda.ExternalStream(stream),
    so it is ordered after the finalize kernel and before the read-back whatever stream torch itself
    considers current -- a context on a private stream and one on torch's default stream both work."""
    import torch
    import torch.distributed as dist

    def hook(ptr, count, stream):
        assert count == NSUMS
        rec = device_doubles(ptr, count, device_index)
        s = int(stream or 0)
        if s == int(torch.cuda.current_stream(device_index).cuda_stream):
            dist.all_reduce(rec, op=dist.ReduceOp.SUM, group=group)
        else:
            with torch.cuda.stream(torch.cuda.ExternalStream(s, device=torch.device("cuda", device_index))):
                dist.all_reduce(rec, op=dist.ReduceOp.SUM, group=group)
        return 0
    return hook
