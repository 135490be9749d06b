"""The RCCL leg of the multi-GPU path on ONE GPU: a world_size-1 "nccl" group exercises the real hook
(device record wrapped without a copy, all-reduce on the context's stream, host read after it).
The 8-GPU run itself is the driver's; the N>1 arithmetic is covered on CPU in test_dist_gloo.py."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_allreduce_hook_over_rccl_single_rank():
    import torch
    import torch.distributed as dist
    import pcl_amd
    from pcl_amd import synth
    from pcl_amd.dist import device_doubles, make_allreduce_hook
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        # the zero-copy wrapper really aliases device memory
        buf = torch.arange(32, dtype=torch.float64, device="cuda")
        alias = device_doubles(buf.data_ptr(), 32, 0)
        alias += 1
        assert torch.equal(buf.cpu(), torch.arange(32, dtype=torch.float64) + 1)
        ctx = pcl_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        tgt = synth.gaussian_surface(200_000, synth.TARGET_SEED)
        src = synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()),
                                synth.gaussian_surface(100_000, synth.SOURCE_SEED))
        tree = pcl_amd.KdTree(ctx)
        tree.setInputCloud(tgt)
        ne = pcl_amd.NormalEstimation(ctx)
        ne.setInputCloud(tgt)
        ne.setSearchMethod(tree)
        ne.setKSearch(8)
        ne.compute(want_output=False)
        results = []
        for use_hook in (False, True):
            icp = pcl_amd.IterativeClosestPointWithNormals(ctx)
            icp.setSearchMethodTarget(tree)
            icp.setInputSource(src)
            icp.setMaximumIterations(6)
            icp.setMaxCorrespondenceDistance(0.1)
            if use_hook:
                icp.setAllReduce(make_allreduce_hook(0))
            icp.align()
            sums = icp.iterate(np.eye(4, dtype=np.float32), max_dist=0.1)
            results.append((icp.getFinalTransformation().copy(), icp.nr_iterations_, sums.copy()))
        # a 1-rank sum is the identity: bit-identical results with and without the collective
        assert np.array_equal(results[0][0], results[1][0]) and results[0][1] == results[1][1]
        assert np.array_equal(results[0][2], results[1][2])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_target_on_device_equals_single_index(world):
    # Target sharding (kd slabs + halo, include/pclhip.h "target sharding") with the real kernels: the `world`
    # ranks are played one after the other on this GPU.  Over all ranks the correspondences of every iteration
    # are those of a single index over the whole target, bit for bit, and the per-rank records add up to the
    # single-index record (their normals are exact: the halo is checked for that).
    import pcl_amd
    from pcl_amd import synth
    from pcl_amd.dist import ShardedTarget, partition_slabs
    ctx = pcl_amd.Context(0)
    n = 300_000
    tgt, src, _ = synth.icp_pair(n)
    tree = pcl_amd.KdTree(ctx)
    tree.setInputCloud(tgt)
    ne = pcl_amd.NormalEstimation(ctx)
    ne.setInputCloud(tgt)
    ne.setSearchMethod(tree)
    ne.setKSearch(8)
    ne.setViewPoint(0, 0, 10)
    ne.compute(want_output=False)
    one = pcl_amd.IterativeClosestPointWithNormals(ctx)
    one.setSearchMethodTarget(tree, True)
    one.setInputSource(src)
    one.reset()
    regions = partition_slabs(tgt, world)
    shards = []
    for r in range(world):
        st = ShardedTarget(ctx, tgt, r, world, 0.1, k_normals=8, viewpoint=(0, 0, 10), regions=regions)
        assert st.normals_exact and st.tree.size() < n and st.margin > 0.1
        icp = pcl_amd.IterativeClosestPointWithNormals(ctx)
        icp.setSearchMethodTarget(st.tree, True)
        icp.setInputSource(src)          # every rank holds the whole source ...
        icp.setRegion(st.region)         # ... and serves the points whose current position it owns
        icp.reset()
        shards.append((st, icp))
    assert sum(st.tree.size() for st, _ in shards) >= n      # slabs + halos cover the cloud
    T = np.eye(4, dtype=np.float32)
    for it in range(4):
        ref = one.iterate(T, max_dist=0.1)
        want = one.fetchCorrespondences()
        total = np.zeros_like(ref)
        got = []
        for st, icp in shards:
            sums = icp.iterate(T, max_dist=0.1)
            total += sums
            got.append(icp.fetchCorrespondences())
        q = np.concatenate([g[0] for g in got])
        order = np.argsort(q, kind="stable")
        assert np.array_equal(q[order], want[0]), it                                        # every query once
        assert np.array_equal(np.concatenate([g[1] for g in got])[order], want[1]), it      # ORIGINAL target indices
        assert np.array_equal(np.concatenate([g[2] for g in got])[order].view(np.uint32), want[2].view(np.uint32)), it
        assert total[28] == ref[28] == n
        assert min(g[0].size for g in got) > 0                                               # every rank had work
        assert np.allclose(total[:28], ref[:28], rtol=1e-10, atol=1e-12), it
        T = one.solve(ref)
    # getFitnessScore under target sharding: a rank scores the points it owns under the given transform (against its slab
    # + halo), so the ranks' (sum, count) pairs add up to the single-index score -- what the all-reduce of a real
    # multi-rank run sums (here the ranks are played one after the other, so the sum is taken by hand)
    for max_range in (1e-6, 0.01):
        want = one.getFitnessScore(max_range, transform=T)
        nr_want = one.fitness_points
        parts = []
        for st, icp in shards:
            sc = icp.getFitnessScore(max_range, transform=T)
            parts.append((sc * icp.fitness_points if icp.fitness_points else 0.0, icp.fitness_points))
        assert sum(p[1] for p in parts) == nr_want and nr_want > 0
        assert abs(sum(p[0] for p in parts) / nr_want - want) <= 1e-12 * want


def test_native_comm_and_region_in_the_device_loop():
    # one rank, all of space as its region, the C-side RCCL all-reduce in every iteration: the device-driven
    # loop must produce exactly what the plain loop produces
    import pcl_amd
    from pcl_amd import synth
    ctx = pcl_amd.Context(0)
    tgt, src, _ = synth.icp_pair(100_000)
    res = []
    for sharded in (False, True):
        icp = pcl_amd.IterativeClosestPoint(ctx)
        icp.setInputTarget(tgt)
        icp.setInputSource(src)
        icp.setMaximumIterations(15)
        icp.setMaxCorrespondenceDistance(0.1)
        if sharded:
            comm = pcl_amd.Communicator(ctx, 0, 1, pcl_amd.Communicator.unique_id())
            icp.setCommunicator(comm)
            icp.setRegion([-np.inf] * 3 + [np.inf] * 3)
        icp.align()
        steps = icp.runSteps(4)
        res.append((icp.getFinalTransformation().copy(), icp.nr_iterations_, [s["num_correspondences"] for s in steps]))
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1:] == res[1][1:]


def test_sharded_bench_path_single_slab_runs_whole_alignments():
    # what bench.py --config 5 does on one GPU: one slab (all of space), the source routed through the region
    # test, the device-driven loop -- the iterations must be those of the plain registration
    import pcl_amd
    from pcl_amd import synth
    from pcl_amd.dist import ShardedTarget
    ctx = pcl_amd.Context(0)
    tgt, src, _ = synth.icp_pair(200_000)
    st = ShardedTarget(ctx, tgt, 0, 1, 0.1, k_normals=8, viewpoint=(0, 0, 10))
    assert st.tree.size() == len(tgt) and np.all(np.isinf(st.region))
    res = []
    for sharded in (True, False):
        icp = pcl_amd.IterativeClosestPointWithNormals(ctx)
        icp.setSearchMethodTarget(st.tree, True)
        icp.setInputSource(src)
        icp.setMaximumIterations(20)
        icp.setMaxCorrespondenceDistance(0.1)
        icp.setTransformationEpsilon(1e-10)
        if sharded:
            icp.setRegion(st.region)
        steps = icp.runSteps(8)
        res.append([(s["iteration"], s["state"], s["num_correspondences"], s["alignment_ended"]) for s in steps])
    assert res[0] == res[1]
    assert [r[0] for r in res[0][:3]] == [1, 2, 3] and res[0][0][2] == len(src)


def test_served_group_lists_equal_the_full_pass(tmp_path):
    # Under target sharding the device-driven loop walks only the 64-point groups of the source whose box touches the
    # rank's region, and brings a group that was skipped for some launches up to date from the transforms it missed.
    # The same run with the option "served_groups" = 0 walks the whole source every launch: the served correspondences must be the
    # same lists -- query, match AND float distance, which only come out equal if every replayed position is the full
    # pass's bit for bit --, the step records the same counts and iterations, the 4x4 equal up to the summation order.
    import subprocess
    import sys
    # (First hardware run: round 4's first GPU call, profiles/r04_first_call.txt; on by default since.)
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "owned_groups_worker.py")
    outs = []
    for owned in ("1", "0"):
        out = str(tmp_path / ("owned%s.npz" % owned))
        r = subprocess.run([sys.executable, worker, out, "90000", owned], capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(np.load(out))
    a, b = outs
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        if k.endswith(("_q", "_m", "_q2", "_m2", "_q3", "_m3", "_its", "_counts", "_iterations")):
            assert np.array_equal(a[k], b[k]), k
        elif k.endswith(("_d", "_d2", "_d3")):
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
        elif k.endswith("_T"):
            assert np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max() < 2e-6, k
        elif k.endswith("_mse"):
            assert np.allclose(a[k], b[k], rtol=1e-9, atol=1e-18), k
    # the scenario does what it is for: regions that serve a part of the cloud, and a cloud that moves through them
    assert 0 < len(a["strip_plane_q"]) < 90000 // 4 and 0 < len(a["corner_plane_q"]) < 90000 // 4
    assert len(a["all_plane_q"]) == 90000
    assert a["strip_plane_counts"][0] != a["strip_plane_counts"][2] and a["corner_point_counts"][0] != a["corner_point_counts"][5]


class _DeviceCopy:
    """a numpy array copied into device memory through the HIP runtime the library itself uses (the emulation's on the CPU
    tier, libamdhip64 on the GPU box) -- no torch in between"""

    def __init__(self, a):
        import ctypes as C
        from pcl_amd import _lib
        lib = _lib.load()
        self.rt = lib if hasattr(lib, "hipMalloc") and os.environ.get("PCLHIP_ALLOW_WAVESIM") == "1" else C.CDLL("libamdhip64.so")
        self.a = np.ascontiguousarray(a)
        self.ptr = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(self.ptr), C.c_size_t(max(self.a.nbytes, 16))) == 0
        if self.a.nbytes:
            assert self.rt.hipMemcpy(self.ptr, C.c_void_p(self.a.ctypes.data), C.c_size_t(self.a.nbytes), 1) == 0

    def free(self):
        self.rt.hipFree(self.ptr)


@pytest.mark.parametrize("kind", ["surface", "lattice", "few", "nan", "line"])
def test_device_partition_and_selection_equal_the_host_code(kind):
    # shard_dev.hip (clouds in device memory: radix selection of the cuts, flag / scan / scatter of the halo) against
    # shard.cpp (host clouds: nth_element, filter): the same regions bit for bit and the same ascending index lists
    import ctypes as C
    import pcl_amd
    from pcl_amd import _lib, synth
    pcl_amd.Context(0)
    lib = _lib.load()
    rng = np.random.default_rng(17)
    if kind == "surface":
        cloud = synth.gaussian_surface(200_003, synth.TARGET_SEED)
    elif kind == "lattice":                      # many equal coordinates: ties at every cut, -0.0 next to +0.0
        cloud = np.ones((50_000, 4), np.float32)
        cloud[:, :3] = rng.integers(-3, 4, (50_000, 3)).astype(np.float32)
        cloud[::7, 0] = -0.0
    elif kind == "few":                          # fewer points than slabs
        cloud = np.ones((5, 4), np.float32)
        cloud[:, :3] = rng.normal(size=(5, 3)).astype(np.float32)
    elif kind == "nan":
        cloud = np.ones((30_000, 4), np.float32)
        cloud[:, :3] = rng.normal(size=(30_000, 3)).astype(np.float32)
        cloud[rng.integers(0, 30_000, 2_000), rng.integers(0, 3, 2_000)] = np.nan
        cloud[100, 1] = np.inf
    else:                                        # a line: two axes without extent
        cloud = np.ones((20_000, 4), np.float32)
        cloud[:, :3] = 0.25
        cloud[:, 1] = rng.uniform(-5, 5, 20_000).astype(np.float32)
    dev = _DeviceCopy(cloud)
    try:
        n, stride = len(cloud), 16
        for slabs in (1, 2, 3, 5, 8):
            host = np.zeros((slabs, 6), np.float32)
            devr = np.zeros((slabs, 6), np.float32)
            fp = C.POINTER(C.c_float)
            _lib.check(lib.pclhip_partition_slabs(C.c_void_p(cloud.ctypes.data), stride, n, slabs, host.ctypes.data_as(fp)))
            _lib.check(lib.pclhip_partition_slabs(dev.ptr, stride, n, slabs, devr.ctypes.data_as(fp)))
            assert np.array_equal(host.view(np.uint32), devr.view(np.uint32)), (kind, slabs, host, devr)
            for g in range(slabs):
                for margin in (0.0, 0.07):
                    lists = []
                    for ptr in (C.c_void_p(cloud.ctypes.data), dev.ptr):
                        cnt = C.c_uint64(0)
                        reg = host[g].ctypes.data_as(fp)
                        st = lib.pclhip_select_region(ptr, stride, n, reg, margin, None, 0, C.byref(cnt))
                        assert st in (0, -5)
                        out = np.empty(int(cnt.value), np.int32)
                        if len(out):
                            _lib.check(lib.pclhip_select_region(ptr, stride, n, reg, margin, C.c_void_p(out.ctypes.data),
                                                                len(out), C.byref(cnt)))
                        lists.append(out)
                    assert np.array_equal(lists[0], lists[1]), (kind, slabs, g, margin)
    finally:
        dev.free()
