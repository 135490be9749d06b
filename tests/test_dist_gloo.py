"""N>1 path on CPU (gloo, world_size 2): source slabs are disjoint and complete, the per-rank
reduction records sum to the single-process record, and every rank solves the same transform.
The device kernel is stood in for by the oracle's accumulation (tests may use the oracle); the
all-reduce + host solve are the product's own code (pcl_amd.dist + pclhip_solve_transformation)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pcl_amd import _lib, synth
from pcl_amd.dist import reduce_record_host, shard_range


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 64, 1000, 10_000_001):
        for w in (1, 2, 3, 8):
            got = [shard_range(n, r, w) for r in range(w)]
            assert sum(c for _, c in got) == n
            pos = 0
            for s, c in got:
                assert s == pos
                pos += c
            assert max(c for _, c in got) - min(c for _, c in got) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_record(tgt, nrm, src, mode):
    from oracle import pcl_oracle as orc
    tree = orc.KdTree(tgt)
    q, m, d = tree.correspondences(src, 0.1, nthreads=2)
    rec = np.zeros(_lib.NSUMS)
    if mode == 1:
        _, s27, used = orc.lls_point_to_plane(src, tgt, nrm, q, m)
        rec[:27] = s27
    elif mode == 2:  # symmetric objective: the source carries normals too (here: the normal of its match, rotated back)
        src_nrm = np.ascontiguousarray(nrm[np.clip(orc.KdTree(tgt).knn(src, 1)[0][:, 0], 0, len(tgt) - 1), :3], np.float32)
        _, s27, used = orc.lls_symmetric(src, src_nrm, tgt, nrm, q, m)
        rec[:27] = s27
    else:
        s, t = src[q, :3].astype(np.float64), tgt[m, :3].astype(np.float64)
        rec[0:3] = s.sum(0)
        rec[3:6] = t.sum(0)
        rec[6:15] = (t.T @ s).reshape(9)
    rec[27] = d.astype(np.float64).sum()
    rec[28] = len(q)
    return rec


def _solve(rec, mode):
    T = np.zeros(16, np.float32)
    assert _lib.load().pclhip_solve_transformation(rec.ctypes.data_as(C.POINTER(C.c_double)), mode,
                                                   T.ctypes.data_as(C.POINTER(C.c_float))) == 0
    return T.reshape(4, 4)


def _worker(rank, world, port, n, mode, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pcl_oracle as orc
        tgt = synth.gaussian_surface(n, synth.TARGET_SEED)
        nrm = orc.KdTree(tgt).normals(tgt, 8, viewpoint=(0, 0, 10), nthreads=2)[0]
        start, count = shard_range(n, rank, world)
        src = synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()),
                                synth.gaussian_surface(count, synth.SOURCE_SEED, start=start))
        rec = reduce_record_host(_local_record(tgt, nrm, src, mode))
        out[rank] = (rec, _solve(rec, mode))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_two_rank_allreduce_matches_single_process(mode):
    n, world = 20000, 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, mode, out), nprocs=world, join=True)
    from oracle import pcl_oracle as orc
    tgt = synth.gaussian_surface(n, synth.TARGET_SEED)
    nrm = orc.KdTree(tgt).normals(tgt, 8, viewpoint=(0, 0, 10), nthreads=2)[0]
    src = synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface(n, synth.SOURCE_SEED))
    ref = _local_record(tgt, nrm, src, mode)
    for r in range(world):
        rec, T = out[r]
        assert rec[28] == ref[28]
        assert np.allclose(rec, ref, rtol=1e-11, atol=1e-12)
        assert np.array_equal(T, out[0][1])               # every rank solves the same system
        assert np.abs(T - _solve(ref, mode)).max() < 1e-7  # and it is the single-process answer


# ---- target sharding (kd slabs + halo, pcl_amd/csrc/shard.cpp): exact by construction -------------------
def _sharded_correspondences(tgt, cur, regions, rank, max_dist):
    """What rank `rank` emits: the points it owns (the kernel's x >= lo && x < hi test) searched in ITS share of
    the target (slab + halo); the device search is stood in for by the oracle, everything else is product code."""
    from oracle import pcl_oracle as orc
    from pcl_amd.dist import region_owner, select_region
    idx = select_region(tgt, regions[rank], max_dist)
    mine = np.nonzero(region_owner(regions, cur) == rank)[0]
    if len(idx) == 0 or len(mine) == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32), len(idx)
    q, m, d = orc.KdTree(np.ascontiguousarray(tgt[idx])).correspondences(np.ascontiguousarray(cur[mine]), max_dist, nthreads=2)
    return mine[q], idx[m], d, len(idx)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_target_equals_single_index(world):
    from oracle import pcl_oracle as orc
    from pcl_amd.dist import partition_slabs, region_owner
    n = 30000
    tgt = synth.gaussian_surface(n, synth.TARGET_SEED)
    tgt[17] = np.nan                                   # dropped by every index, owned by nobody
    tgt[100:120] = tgt[200:220]                        # exact duplicates: ties go to the lower original index
    src = synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface(n, synth.SOURCE_SEED))
    regions = partition_slabs(tgt, world)
    own_t = region_owner(regions, tgt)
    cnt = np.bincount(own_t[own_t >= 0], minlength=world)
    assert cnt.sum() == n - 1 and cnt.max() - cnt.min() <= 25 and own_t[17] == -1     # equal counts (duplicates aside)
    # the regions tile space: every finite point of ANY cloud has exactly one owner
    probe = np.random.default_rng(0).uniform(-3, 3, (5000, 3)).astype(np.float32)
    assert (region_owner(regions, probe) >= 0).all()
    tree = orc.KdTree(tgt)
    for max_dist, T in ((0.1, np.eye(4)), (0.02, np.eye(4)), (0.1, synth.ground_truth_transform())):
        cur = synth.apply_rigid(T, src)
        want = tree.correspondences(cur, max_dist, nthreads=2)
        got_q, got_m, got_d, halo = [], [], [], 0
        for r in range(world):
            q, m, d, h = _sharded_correspondences(tgt, cur, regions, r, max_dist)
            got_q.append(q); got_m.append(m); got_d.append(d)
            halo += h
        order = np.argsort(np.concatenate(got_q), kind="stable")
        q = np.concatenate(got_q)[order]
        assert np.array_equal(q, want[0])                                        # every query exactly once
        assert np.array_equal(np.concatenate(got_m)[order], want[1])             # the same match, ORIGINAL indices
        assert np.array_equal(np.concatenate(got_d)[order].view(np.uint32), want[2].view(np.uint32))
        assert halo >= n - 1                                                      # slabs + halos cover the cloud


@pytest.mark.parametrize("cloud,n_slabs", [
    (np.array([[-2.5, 1.0, 0.5, 1.0]], np.float32), 4),                                  # one point, negative x, four slabs
    (np.array([[7.0, -1.0, 2.0, 1.0], [7.0, -1.0, 2.0, 1.0]], np.float32), 3),            # fewer distinct points than slabs
    (np.tile(np.array([[0.25, 0.5, -0.75, 1.0]], np.float32), (64, 1)), 8),               # every coordinate duplicated
    (np.c_[np.repeat(np.array([-3.0, -3.0, 5.0], np.float32), 50), np.zeros(150, np.float32), np.zeros(150, np.float32),
           np.ones(150, np.float32)].astype(np.float32), 5),                             # heavy ties at the cuts
    (np.zeros((0, 4), np.float32), 3),                                                    # an empty cloud
])
def test_partition_slabs_tiles_space_whatever_the_cloud(cloud, n_slabs):
    # the regions tile space with exactly one owner per point even when there is nothing to cut: no region with lo > hi,
    # no overlap (the C owner test takes the first match, the Python one the last: they have to agree)
    from pcl_amd.dist import partition_slabs, region_owner
    regions = partition_slabs(cloud, n_slabs)
    assert regions.shape == (n_slabs, 6) and (regions[:, :3] <= regions[:, 3:]).all()
    probe = np.random.default_rng(1).uniform(-10, 10, (4000, 3)).astype(np.float32)
    probe = np.concatenate([probe, cloud[:, :3]]) if len(cloud) else probe
    inside = np.stack([np.all((probe >= regions[g, :3]) & (probe < regions[g, 3:]), axis=1) for g in range(n_slabs)])
    assert (inside.sum(0) == 1).all()                       # exactly one owner, first match == last match
    assert (region_owner(regions, probe) >= 0).all()
    if len(cloud):
        own = region_owner(regions, cloud)
        assert (own >= 0).all()


def _shard_worker(rank, world, port, n, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pcl_oracle as orc
        from pcl_amd.dist import partition_slabs
        tgt = synth.gaussian_surface(n, synth.TARGET_SEED)
        src = synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface(n, synth.SOURCE_SEED))
        # rank 0 partitions, everybody uses ITS answer (as a real job would broadcast the 6 floats per rank)
        box = [partition_slabs(tgt, world) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        regions = box[0]
        nrm = orc.KdTree(tgt).normals(tgt, 8, viewpoint=(0, 0, 10), nthreads=2)[0]
        q, m, d, _ = _sharded_correspondences(tgt, src, regions, rank, 0.1)
        rec = np.zeros(_lib.NSUMS)
        _, s27, used = orc.lls_point_to_plane(src, tgt, nrm, q.astype(np.int32), m)
        rec[:27] = s27
        rec[27] = d.astype(np.float64).sum()
        rec[28] = len(q)
        rec = reduce_record_host(rec)
        out[rank] = (q, m, rec, _solve(rec, 1))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_target_matches_single_process():
    n, world = 20000, 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_shard_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
    from oracle import pcl_oracle as orc
    tgt = synth.gaussian_surface(n, synth.TARGET_SEED)
    nrm = orc.KdTree(tgt).normals(tgt, 8, viewpoint=(0, 0, 10), nthreads=2)[0]
    src = synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface(n, synth.SOURCE_SEED))
    ref = _local_record(tgt, nrm, src, 1)
    want = orc.KdTree(tgt).correspondences(src, 0.1, nthreads=2)
    q = np.concatenate([out[r][0] for r in range(world)])
    m = np.concatenate([out[r][1] for r in range(world)])
    order = np.argsort(q, kind="stable")
    assert np.array_equal(q[order], want[0]) and np.array_equal(m[order], want[1])
    assert len(out[0][0]) > 0 and len(out[1][0]) > 0                      # both ranks served queries
    for r in range(world):
        rec, T = out[r][2], out[r][3]
        assert rec[28] == ref[28] == n
        assert np.allclose(rec, ref, rtol=1e-11, atol=1e-12)
        assert np.array_equal(T, out[0][3])
        assert np.abs(T - _solve(ref, 1)).max() < 1e-7


def _fallback_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pcl_amd.api as api
        from pcl_amd.dist import attach_collective, native_communicator

        # rank 1 cannot bind RCCL: NO rank may go on to create (and then wait inside) the communicator
        def broken():
            raise OSError("librccl.so: cannot open shared object file")

        def working():
            return bytes(_lib.COMM_ID_BYTES)

        created = []

        class FakeComm:
            def __init__(self, *a):
                created.append(a)

        api.Communicator.unique_id = staticmethod(broken if rank == 1 else working)
        real = api.Communicator.__init__
        api.Communicator.__init__ = lambda self, *a: created.append(a)
        try:
            comm = native_communicator(object(), rank, world)
        finally:
            api.Communicator.__init__ = real
        # ... and a job without a native communicator must not silently run unreduced under a backend with no device collective
        refused = False
        try:
            attach_collective(object(), None, 0, world)
        except RuntimeError:
            refused = True
        out[rank] = (comm is None, len(created), refused)
    finally:
        dist.destroy_process_group()


def test_ranks_agree_to_fall_back_when_one_cannot_bind_rccl():
    """pcl_amd.dist.native_communicator: one rank failing to bind RCCL makes EVERY rank return None before anyone joins the
    communicator (bench.py then sums the record through torch.distributed: attach_collective)."""
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_fallback_worker, args=(world, port, out), nprocs=world, join=True)
        assert [out[r] for r in range(world)] == [(True, 0, True)] * world
