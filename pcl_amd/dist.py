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


# ---- bring-up of an N-rank job: ONE implementation, used by bench.py (backend "nccl" = RCCL, one rank per GPU) and by the
# multi-process CPU tests (tests/wavesim/two_rank_worker.py: backend "gloo", one emulated device per process), so that the
# control flow the driver's 8-GPU run depends on -- rendezvous, id broadcast, the second (native) communicator next to
# torch's, the fences around the timed region -- has run with more than one rank before it ever meets hardware.
def init_ranks(backend="nccl"):
    """(rank, local_rank, world) from the launcher's environment (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_*); world > 1: the default torch.distributed group is created (device-bound for "nccl")."""
    import os
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")   # the container hostname may not resolve
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def _all_ranks_ok(ok):
    """True when `ok` holds on EVERY rank (a MIN all-reduce over the default torch group)."""
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def native_communicator(ctx, rank, world):
    """The library's own RCCL communicator (pclhip_comm_*: its all-reduce is issued from C on the context's stream): rank 0
    draws the 128-byte id, the default torch.distributed group broadcasts it once (a device tensor under "nccl", a host tensor
    under "gloo"), every rank joins.  None for a single rank -- and None, on EVERY rank, when some rank could not bind RCCL or
    join (the ranks agree on that through the torch group before anyone depends on the communicator): attach_collective()
    then sums the record through torch.distributed on the context's stream instead, so an N-rank job still runs."""
    if world <= 1:
        return None
    import sys
    import torch
    import torch.distributed as dist
    from . import _lib
    from .api import Communicator
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    uid, ok = None, True
    try:
        uid = Communicator.unique_id()   # every rank: binds RCCL's entry points (rank 0's id is the one that travels)
    except Exception as e:  # noqa: BLE001 -- whatever went wrong, the job falls back as a whole
        print("pcl_amd.dist: rank %d cannot bind RCCL for the native communicator (%s)" % (rank, e), file=sys.stderr)
        ok = False
    if not _all_ranks_ok(ok):
        return None
    t = torch.frombuffer(bytearray(uid), dtype=torch.uint8).to(dev)
    dist.broadcast(t, 0)
    comm = None
    try:
        comm = Communicator(ctx, rank, world, bytes(t.cpu().numpy().tobytes()))
    except Exception as e:  # noqa: BLE001
        print("pcl_amd.dist: rank %d could not join the native communicator (%s)" % (rank, e), file=sys.stderr)
    if not _all_ranks_ok(comm is not None):
        if comm is not None:
            comm._release()
        return None
    return comm


def attach_collective(icp, comm, local_rank, world):
    """Give `icp` its per-iteration all-reduce: the native communicator when there is one, else (world > 1) torch.distributed's
    all-reduce enqueued on the context's stream (make_allreduce_hook).  Returns "native", "torch" or None (single rank)."""
    if comm is not None:
        icp.setCommunicator(comm)
        return "native"
    if world > 1:
        import torch.distributed as dist
        if dist.get_backend() == "nccl":
            icp.setAllReduce(make_allreduce_hook(local_rank))
        else:   # gloo: host tensors (the CPU tests)
            raise RuntimeError("no native communicator and no device collective under backend %r" % dist.get_backend())
        return "torch"
    return None


def make_fence(ctx, world):
    """fence(): everything the context and torch have queued is done on EVERY rank (barrier between two device-wide waits):
    what brackets a timed region."""
    import torch
    import torch.distributed as dist
    on_gpu = torch.cuda.is_available()

    def fence():
        ctx.synchronize()
        if on_gpu:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()
    return fence


def max_over_ranks(value, world):
    """the slowest rank's figure (bench.py: timed seconds), identical on every rank"""
    if world <= 1:
        return float(value)
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def timed_steps(icp, steps, warmup, fence, world):
    """bench.py's timed region: `warmup` untimed steps, then exactly `steps` steps (whole alignments queued back to back on
    the device, restarting there) bracketed by fence() on both sides; (step records, seconds = the MAX over the ranks)."""
    import time
    if warmup > 0:
        icp.runSteps(warmup)
    fence()
    t0 = time.perf_counter()
    records = icp.runSteps(steps)
    fence()
    return records, max_over_ranks(time.perf_counter() - t0, world)


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
    """Hook for IterativeClosestPoint.setAllReduce: sums a DEVICE buffer of doubles (the 32-double record of an iteration,
    or the 2048-bin histogram of a rejector's selection pass) in place over RCCL, ON THE
    STREAM THE C SIDE PASSES (the context's stream, where the record was just produced and from which the
    host copy is issued afterwards).  The collective is enqueued under torch.cuda.ExternalStream(stream),
    so it is ordered after the finalize kernel and before the read-back whatever stream torch itself
    considers current -- a context on a private stream and one on torch's default stream both work."""
    import torch
    import torch.distributed as dist

    def hook(ptr, count, stream):
        rec = device_doubles(ptr, count, device_index)
        s = int(stream or 0)
        if s == int(torch.cuda.current_stream(device_index).cuda_stream):
            dist.all_reduce(rec, op=dist.ReduceOp.SUM, group=group)
        else:
            with torch.cuda.stream(torch.cuda.ExternalStream(s, device=torch.device("cuda", device_index))):
                dist.all_reduce(rec, op=dist.ReduceOp.SUM, group=group)
        return 0
    return hook


# ---- target sharding: kd slabs + halo over the ranks -----------------------------------
# The partition / selection / ownership logic is host code of libpclhip.so (pcl_amd/csrc/shard.cpp): it runs
# without a GPU, which is what the multi-process CPU tests exercise.
def _cloud_arg(cloud):
    import ctypes as C
    if type(cloud).__module__.startswith("torch"):
        c = cloud.contiguous()
        return C.c_void_p(c.data_ptr()), c.shape[1] * 4, c.shape[0], c
    c = np.ascontiguousarray(cloud, np.float32)
    return C.c_void_p(c.ctypes.data), c.shape[1] * 4, c.shape[0], c


def partition_slabs(cloud, n_slabs):
    """(n_slabs, 6) float32 regions (lo.xyz, hi.xyz): kd cells of equal point count that tile space."""
    import ctypes as C
    from . import _lib
    ptr, stride, n, keep = _cloud_arg(cloud)
    reg = np.zeros((int(n_slabs), 6), np.float32)
    _lib.check(_lib.load().pclhip_partition_slabs(ptr, stride, n, int(n_slabs), reg.ctypes.data_as(C.POINTER(C.c_float))))
    return reg


def select_region(cloud, region, margin):
    """ascending int32 indices of the finite points of `cloud` inside `region` dilated by `margin`"""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    ptr, stride, n, keep = _cloud_arg(cloud)
    reg = np.ascontiguousarray(region, np.float32).reshape(6)
    cnt = C.c_uint64(0)
    rp = reg.ctypes.data_as(C.POINTER(C.c_float))
    st = lib.pclhip_select_region(ptr, stride, n, rp, float(margin), None, 0, C.byref(cnt))
    if st not in (0, -5):
        _lib.check(st)
    out = np.empty(int(cnt.value), np.int32)
    if len(out):
        _lib.check(lib.pclhip_select_region(ptr, stride, n, rp, float(margin), C.c_void_p(out.ctypes.data), len(out),
                                            C.byref(cnt)))
    return out


def region_owner(regions, points):
    """owner slab of every point (x >= lo && x < hi per axis, the kernel's test); -1 for non-finite points"""
    r = np.asarray(regions, np.float32).reshape(-1, 6)
    p = np.asarray(points, np.float32)[:, :3]
    own = np.full(len(p), -1, np.int32)
    for g in range(len(r)):
        inside = np.all((p >= r[g, :3]) & (p < r[g, 3:]), axis=1)
        own[inside] = g
    return own


class ShardedTarget:
    """This rank's share of a target cloud spread over `world` GPUs: its kd slab plus the halo, indexed on the
    device (results carry indices into the ORIGINAL cloud), with normals that are exact wherever a query this
    rank serves can be matched (checked, not assumed).

      margin = max_correspondence_distance + normals_margin
    The first term makes correspondences exact, the second leaves every matchable point all of its k
    neighbours inside the halo; `normals_margin` None picks 4x the k-th neighbour distance measured on the slab."""

    def __init__(self, ctx, target, rank, world, max_correspondence_distance, k_normals=0, viewpoint=(0.0, 0.0, 0.0),
                 normals_margin=None, regions=None):
        from . import api
        import time
        tm = self.timings = {}   # seconds per stage of the setup (bench.py --config 5 reports them)

        def lap(name, t0):
            ctx.synchronize()
            tm[name] = round(time.perf_counter() - t0, 4)
        t0 = time.perf_counter()
        self.regions = partition_slabs(target, world) if regions is None else np.asarray(regions, np.float32).reshape(world, 6)
        lap("partition", t0)
        self.region = self.regions[rank].copy()
        md = float(max_correspondence_distance)
        self.tree = api.KdTree(ctx)
        extra = 0.0 if k_normals <= 0 else (normals_margin if normals_margin is not None else None)
        if extra is None:  # measure the neighbourhood size on the bare slab first
            t0 = time.perf_counter()
            probe = api.KdTree(ctx)
            probe.setInputCloud(target, select_region(target, self.region, 0.0))
            extra = 4.0 * probe.kthDistanceMax(k_normals)
            probe._free()
            lap("probe", t0)
        self.margin = md + float(extra)
        t0 = time.perf_counter()
        self.indices = select_region(target, self.region, self.margin)
        lap("select", t0)
        t0 = time.perf_counter()
        self.tree.setInputCloud(target, self.indices)
        lap("index", t0)
        self.normals_exact = None
        if k_normals > 0:
            ne = api.NormalEstimation(ctx)
            ne.setInputCloud(target)
            ne.setSearchMethod(self.tree)
            ne.setKSearch(k_normals)
            ne.setViewPoint(*viewpoint)
            ne.tree = self.tree
            t0 = time.perf_counter()
            self._compute_normals(ne)
            lap("normals", t0)
            t0 = time.perf_counter()
            # every target point a query of this region can be matched to lies within md of the region
            box = np.concatenate([self.region[:3] - np.float32(md * 1.00002), self.region[3:] + np.float32(md * 1.00002)])
            self.kth = self.tree.kthDistanceMax(k_normals, box)
            lap("halo_check", t0)
            self.normals_exact = self.kth <= extra
            if not self.normals_exact:
                raise RuntimeError("halo too thin for exact normals: k-th neighbour at %.3g, margin beyond max_dist %.3g"
                                   % (self.kth, extra))

    def _compute_normals(self, ne):
        import ctypes as C
        from . import _lib
        nan = C.c_uint64(0)
        _lib.check(self.tree.lib.pclhip_normals(self.tree.h, ne.k, ne.vp.ctypes.data_as(C.POINTER(C.c_float)), None, 0,
                                                C.byref(nan)), self.tree.ctx.h)
