"""Parity AT THE SIZES THE METRIC IS QUOTED ON (BASELINE.json configs 2, 3, 4): the HIP path through the
C ABI against the CPU oracle, bit for bit where the contract says so.

  config 2   2^20-point clouds, k = 1, point-to-point (TransformationEstimationSVD)
  config 3   10M-point clouds, k = 8 NormalEstimation + point-to-plane ICP
  config 4   config 3's clouds through VoxelGrid(0.01) first, then normals + ICP on the filtered clouds

Reference lines the comparisons follow: registration/include/pcl/registration/impl/
correspondence_estimation.hpp:145-218 (correspondences), impl/icp.hpp:113-268 (loop),
filters/include/pcl/filters/impl/voxel_grid.hpp:597-814, features/include/pcl/features/impl/normal_3d.hpp:48-95.
The deeper tree of the 10M index (levels 3-4, the LDS top-level cache boundary, the start-level shortcut of
seeded iterations) is only exercised at these sizes.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N2 = 1 << 20
N3 = 10_000_000
ICP_KW = dict(max_iterations=20, max_correspondence_distance=0.1, transformation_epsilon=1e-10)


@pytest.fixture(scope="module")
def gpu():
    from conftest import make_context      # PCLHIP_TEST_OPTIONS: tests/conftest.py
    return make_context(0)


@pytest.fixture(scope="module")
def orc():
    from oracle import pcl_oracle
    return pcl_oracle


def frob(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)))


def assert_same_correspondences(got, want, what):
    q, m, d = got
    oq, om, od = want
    assert len(q) == len(oq), (what, len(q), len(oq))
    assert np.array_equal(q, oq), what
    bad = np.nonzero(m != om)[0]
    assert len(bad) == 0, (what, len(bad), q[bad[:5]], m[bad[:5]], om[bad[:5]])
    assert np.array_equal(d.view(np.uint32), od.view(np.uint32)), what


# ------------------------------------------------------------------------------------------------
# config 2
# ------------------------------------------------------------------------------------------------
def test_config2_correspondences_and_svd_bit_exact_at_2e20(gpu, orc):
    import pcl_amd
    tgt, src, T_gt = pcl_amd.synth.icp_pair(N2)
    otree = orc.KdTree(tgt)
    # the oracle's whole alignment, every iteration's transform and match list recorded.  acc_double: the
    # reference sums umeyama's moments in float in Eigen's internal order (not reproducible); the oracle
    # offers the sequential-float order and the exact-arithmetic (double) limit -- the device sums in fp64.
    ref = orc.icp_align(otree, tgt, src, mode=0, record=True, acc_double=1, **ICP_KW)
    assert ref["iterations"] >= 4
    icp = pcl_amd.IterativeClosestPoint(gpu)
    icp.setInputTarget(tgt)
    icp.setInputSource(src)
    icp.reset()
    # (a) every iteration driven by the ORACLE's transform: correspondences bit for bit, all 2^20 of them
    T_prev = np.eye(4, dtype=np.float32)
    cur = src.copy()
    for it in range(min(ref["iterations"], 5)):
        sums = icp.iterate(T_prev, max_dist=0.1)
        cur = orc.transform_cloud(T_prev, cur, order=0)
        want = otree.correspondences(cur, 0.1)
        assert_same_correspondences(icp.fetchCorrespondences(), want, "iteration %d" % it)
        row = ref["per_iter_match"][it]
        assert np.array_equal(want[1], row[row >= 0])            # ... which is what the oracle's own loop saw
        assert int(sums[28]) == len(want[0])
        # double sums on both sides: the closed form agrees to float rounding of the 4x4
        assert np.abs(icp.solve(sums) - ref["per_iter_T"][it]).max() < 2e-6, it
        T_prev = ref["per_iter_T"][it]
    # (b) the free-running alignment: same iteration count, final 4x4 within the 1e-5 contract
    icp.setMaximumIterations(ICP_KW["max_iterations"])
    icp.setMaxCorrespondenceDistance(0.1)
    icp.setTransformationEpsilon(1e-10)
    icp.align()
    assert icp.nr_iterations_ == ref["iterations"]
    assert icp.getConvergenceState() == pcl_amd._lib.CONVERGENCE_STATES[ref["state"]]
    assert frob(icp.getFinalTransformation(), ref["T"]) < 1e-5
    # (c) how far the reference's float sums can be from that: the oracle's sequential-float variant, same
    # inputs -- documented, bounded (weak #10 of the round-1 review): at 2^20 points the float-sum noise of
    # umeyama's moments is what separates the two, not the search
    ref_f = orc.icp_align(otree, tgt, src, mode=0, acc_double=0, **ICP_KW)
    gap = frob(ref_f["T"], ref["T"])
    to_float_order = frob(icp.getFinalTransformation(), ref_f["T"])
    print("config 2: |T_gpu - T_oracle(double sums)|_F = %.3g ; float-sum vs double-sum oracle gap = %.3g ; "
          "|T_gpu - T_oracle(reference-order float sums)|_F = %.3g = %.0f %% of the 1e-5 contract ; iterations %d" %
          (frob(icp.getFinalTransformation(), ref["T"]), gap, to_float_order, 100.0 * to_float_order / 1e-5,
           ref["iterations"]))
    # the contract itself (north_star: 1e-5 Frobenius of the CPU path), against the oracle variant that sums in the
    # reference's own float order -- not only against the double-sum variant the device arithmetic coincides with
    assert ref_f["iterations"] == ref["iterations"]
    assert to_float_order < 1e-5
    assert gap < 1e-5


# ------------------------------------------------------------------------------------------------
# config 3 (and the 10M property checks that need no oracle)
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def clouds10m():
    from pcl_amd import synth
    tgt = synth.gaussian_surface(N3, synth.TARGET_SEED)
    src = synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface(N3, synth.SOURCE_SEED))
    return tgt, src


@pytest.fixture(scope="module")
def tree10m(gpu, clouds10m):
    import torch
    import pcl_amd
    tgt_d = torch.from_numpy(clouds10m[0]).cuda()
    tree = pcl_amd.KdTree(gpu)
    tree.setInputCloud(tgt_d)
    return tree, tgt_d


@pytest.fixture(scope="module")
def otree10m(orc, clouds10m):
    return orc.KdTree(clouds10m[0])


def test_config3_knn8_and_normals_on_every_point_of_the_10m_cloud(gpu, orc, clouds10m, tree10m, otree10m):
    # (rounds 2-3 checked every 10th point; the oracle answers 10M k = 8 queries in seconds on the GPU box's cores)
    import pcl_amd
    tgt, _ = clouds10m
    tree, tgt_d = tree10m
    for a in range(0, N3, 2_500_000):                    # four slices: 80 MB of indices + 80 MB of distances at a time
        q = np.ascontiguousarray(tgt[a:a + 2_500_000])
        gi, gd = tree.nearestKSearch(q, 8)
        oi, od = otree10m.knn(q, 8)
        assert np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32)), a
    ne = pcl_amd.NormalEstimation(gpu)
    ne.setInputCloud(tgt_d)
    ne.setSearchMethod(tree)
    ne.setKSearch(8)
    ne.setViewPoint(0, 0, 10)
    nrm = ne.compute().cpu().numpy()
    assert ne.nan_count == 0
    onrm, nan = otree10m.normals(tgt, 8, viewpoint=(0, 0, 10))
    assert nan == 0
    dots = np.abs(np.sum(nrm[:, :3].astype(np.float64) * onrm[:, :3].astype(np.float64), axis=1))
    assert dots.min() > 1 - 1e-5, dots.min()
    assert np.all(np.sum(nrm[:, :3] * onrm[:, :3], axis=1) > 0)     # same orientation (viewpoint flip)
    assert np.abs(nrm[:, 3] - onrm[:, 3]).max() < 1e-5


def test_standoff_search_beyond_the_old_size_gate_at_20m(gpu, orc):
    # Until round 6 the stand-off search (standoff.hpp) served indices up to 640 MB (12M points); the gate is open now
    # (search.hip: the measurements).  The launch that starts an alignment and two seeded ones at 20M points -- an index of
    # 1.1 GB -- against the oracle: every correspondence, index and float distance.
    import torch
    import pcl_amd
    from pcl_amd import synth
    n = 20_000_000
    tgt = synth.gaussian_surface_device(n, synth.TARGET_SEED)
    src = synth.apply_rigid_device(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface_device(n, synth.SOURCE_SEED))
    torch.cuda.synchronize()
    tree = pcl_amd.KdTree(gpu)
    tree.setInputCloud(tgt)
    tgt_h, src_h = tgt.cpu().numpy(), src.cpu().numpy()
    otree = orc.KdTree(tgt_h)
    icp = pcl_amd.IterativeClosestPoint(gpu)
    icp.setSearchMethodTarget(tree, True)
    icp.setInputSource(src)
    icp.reset()
    T = np.eye(4, dtype=np.float32)
    cur = src_h.copy()
    for it in range(3):
        sums = icp.iterate(T, max_dist=0.1)
        cur = orc.transform_cloud(T, cur, order=0)
        assert_same_correspondences(icp.fetchCorrespondences(), otree.correspondences(cur, 0.1), "20M points, launch %d" % it)
        T = icp.solve(sums)


def test_config3_icp_correspondences_bit_exact_at_10m(gpu, orc, clouds10m, tree10m, otree10m):
    import torch
    import pcl_amd
    tgt, src = clouds10m
    tree, tgt_d = tree10m
    onrm, nan = otree10m.normals(tgt, 8, viewpoint=(0, 0, 10))
    assert nan == 0
    ref = orc.icp_align(otree10m, tgt, src, mode=1, tgt_normals=onrm, record=True, **ICP_KW)
    assert 3 <= ref["iterations"] <= 6
    icp = pcl_amd.IterativeClosestPointWithNormals(gpu)
    icp.setSearchMethodTarget(tree, True)
    icp.setTargetNormals(onrm)                 # same normals on both sides: isolates the search
    icp.setInputSource(torch.from_numpy(src).cuda())
    icp.reset()
    # cold iteration + the seeded ones, driven by the oracle's transforms: every one of the 10M
    # correspondences (index AND float distance) equals the oracle's
    T_prev = np.eye(4, dtype=np.float32)
    cur = src.copy()
    for it in range(ref["iterations"]):          # all of them: the converged ones take the start-level shortcut most often
        sums = icp.iterate(T_prev, max_dist=0.1)
        cur = orc.transform_cloud(T_prev, cur, order=1)
        want = otree10m.correspondences(cur, 0.1)
        assert_same_correspondences(icp.fetchCorrespondences(), want, "iteration %d" % it)
        row = ref["per_iter_match"][it]
        assert np.array_equal(want[1], row[row >= 0])
        assert int(sums[28]) == len(want[0]) == N3
        assert np.abs(icp.solve(sums) - ref["per_iter_T"][it]).max() < 1e-6, it
        T_prev = ref["per_iter_T"][it]
    # free-running alignment with the DEVICE's own normals: iteration count, state, 1e-5 on the 4x4
    ne = pcl_amd.NormalEstimation(gpu)
    ne.setInputCloud(tgt_d)
    ne.setSearchMethod(tree)
    ne.setKSearch(8)
    ne.setViewPoint(0, 0, 10)
    ne.compute(want_output=False)
    icp2 = pcl_amd.IterativeClosestPointWithNormals(gpu)
    icp2.setSearchMethodTarget(tree, True)
    icp2.setInputSource(torch.from_numpy(src).cuda())
    icp2.setMaximumIterations(ICP_KW["max_iterations"])
    icp2.setMaxCorrespondenceDistance(0.1)
    icp2.setTransformationEpsilon(1e-10)
    icp2.align()
    assert icp2.hasConverged() and icp2.nr_iterations_ == ref["iterations"]
    err = frob(icp2.getFinalTransformation(), ref["T"])
    print("config 3: |T_gpu - T_oracle|_F = %.3g, %d iterations, |T - T_gt|_F = %.3g" %
          (err, icp2.nr_iterations_, frob(icp2.getFinalTransformation(), pcl_amd.synth.ground_truth_transform())))
    assert err < 1e-5


def test_full_size_properties_10m(gpu, clouds10m, tree10m):
    # properties that need no oracle, on all 10M points
    import torch
    import pcl_amd
    from pcl_amd import synth
    n = N3
    tgt_h, _ = clouds10m
    tree, tgt = tree10m
    # (1) self-queries: every point finds itself at distance 0 (ties between duplicates -> lower index)
    idx, d2 = tree.nearestKSearch(tgt, 1)
    assert int((d2 != 0).sum()) == 0
    ar = torch.arange(n, device="cuda", dtype=torch.int32)
    moved = idx[:, 0] != ar
    assert bool((idx[:, 0][moved] < ar[moved]).all())            # only duplicates, resolved downwards
    assert bool((tgt[idx[:, 0][moved].long(), :3] == tgt[moved, :3]).all())
    # (2) k = 8: ascending distances, first neighbour is the point itself, no index repeats in a row
    idx8, d8 = tree.nearestKSearch(tgt[:2_000_000], 8)
    assert bool((d8[:, 1:] >= d8[:, :-1]).all()) and int((d8[:, 0] != 0).sum()) == 0
    s = torch.sort(idx8, dim=1).values
    assert int((s[:, 1:] == s[:, :-1]).sum()) == 0
    # (3) a rigidly moved copy of the target: after undoing the motion every source point matches its own
    #     original (or an exact duplicate) within float rounding, all 10M correspondences are kept
    T = synth.ground_truth_transform().astype(np.float32)
    icp = pcl_amd.IterativeClosestPoint(gpu)
    icp.setSearchMethodTarget(tree, True)
    src = torch.from_numpy(synth.apply_rigid(T, tgt_h)).cuda()
    icp.setInputSource(src)
    icp.reset()
    sums = icp.iterate(np.linalg.inv(T.astype(np.float64)).astype(np.float32), max_dist=0.1)
    assert sums[28] == n and sums[27] / n < 1e-12                 # mean squared distance ~ rounding^2
    # (4) idempotence: a second iteration with the identity reproduces the same record bit for bit
    again = icp.iterate(np.eye(4, dtype=np.float32), max_dist=0.1)
    assert np.array_equal(again[27:29], sums[27:29]) and np.array_equal(again[:15], sums[:15])


# ------------------------------------------------------------------------------------------------
# config 4: VoxelGrid(0.01) -> NormalEstimation(k = 8) -> point-to-plane ICP, chained on the device
# ------------------------------------------------------------------------------------------------
def test_config4_voxelgrid_normals_icp_chain_vs_oracle(gpu, orc, clouds10m):
    import torch
    import pcl_amd
    tgt, src = clouds10m
    # device chain
    filt = {}
    for name, cloud in (("tgt", tgt), ("src", src)):
        vg = pcl_amd.VoxelGrid(gpu)
        vg.setInputCloud(torch.from_numpy(cloud).cuda())
        vg.setLeafSize(0.01, 0.01, 0.01)
        filt[name] = vg.filter()
    # oracle chain
    ofilt = {name: orc.voxelgrid(cloud, 0.01)[0] for name, cloud in (("tgt", tgt), ("src", src))}
    for name in ("tgt", "src"):
        got = filt[name].cpu().numpy()
        assert got.shape == ofilt[name].shape and 5e4 < len(got) < 9e4, (name, got.shape)
        assert np.array_equal(got.view(np.uint32), ofilt[name].view(np.uint32)), name   # centroids bit for bit
    tree = pcl_amd.KdTree(gpu)
    tree.setInputCloud(filt["tgt"])
    ne = pcl_amd.NormalEstimation(gpu)
    ne.setInputCloud(filt["tgt"])
    ne.setSearchMethod(tree)
    ne.setKSearch(8)
    ne.setViewPoint(0, 0, 10)
    nrm = ne.compute().cpu().numpy()
    otree = orc.KdTree(ofilt["tgt"])
    onrm, nan = otree.normals(ofilt["tgt"], 8, viewpoint=(0, 0, 10))
    assert nan == 0 and ne.nan_count == 0
    assert np.abs(np.sum(nrm[:, :3].astype(np.float64) * onrm[:, :3], axis=1)).min() > 1 - 1e-5
    icp = pcl_amd.IterativeClosestPointWithNormals(gpu)
    icp.setSearchMethodTarget(tree, True)
    icp.setInputSource(filt["src"])
    icp.setMaximumIterations(ICP_KW["max_iterations"])
    icp.setMaxCorrespondenceDistance(0.1)
    icp.setTransformationEpsilon(1e-10)
    icp.align()
    ref = orc.icp_align(otree, ofilt["tgt"], ofilt["src"], mode=1, tgt_normals=onrm, **ICP_KW)
    err = frob(icp.getFinalTransformation(), ref["T"])
    print("config 4: %d / %d filtered points, |T_gpu - T_oracle|_F = %.3g, %d iterations, |T - T_gt|_F = %.3g" %
          (len(ofilt["tgt"]), len(ofilt["src"]), err, icp.nr_iterations_,
           frob(icp.getFinalTransformation(), pcl_amd.synth.ground_truth_transform())))
    assert icp.nr_iterations_ == ref["iterations"] and icp.hasConverged() == ref["converged"]
    assert err < 1e-5
    assert frob(icp.getFinalTransformation(), pcl_amd.synth.ground_truth_transform()) < 5e-3


def test_config3_device_driven_loop_correspondences_bit_exact_at_10m(gpu, orc, clouds10m, tree10m, otree10m):
    # The DEVICE-DRIVEN loop (icp_search_dual_kernel: the stand-off body inside the first launch of an alignment, the
    # seeded body after it, transforms handed from launch to launch in device memory) against the oracle, iteration by
    # iteration, at the bench's size: an alignment capped at K iterations leaves iteration K's matches behind; its
    # incremental transforms (bit for bit the same from run to run: fixed-order fp64 sums) move the oracle's copy of
    # the cloud with the device's own arithmetic, and every one of the 10M correspondences of every iteration -- index
    # AND float distance -- must be the oracle's.  (the emulation's 3M-point check, on hardware.)
    import torch
    import pcl_amd
    tgt, src = clouds10m
    tree, tgt_d = tree10m
    ne = pcl_amd.NormalEstimation(gpu)
    ne.setInputCloud(tgt_d)
    ne.setSearchMethod(tree)
    ne.setKSearch(8)
    ne.setViewPoint(0, 0, 10)
    ne.compute(want_output=False)
    gpu.setOption("icp_lookahead", 0)          # no launch queued beyond the last iteration of a capped alignment
    try:
        src_d = torch.from_numpy(src).cuda()
        cur = src.copy()
        full = None
        for K in range(1, 7):
            icp = pcl_amd.IterativeClosestPointWithNormals(gpu)   # a fresh object: the criteria keep their memory across align() calls
            icp.setSearchMethodTarget(tree, True)
            icp.setInputSource(src_d)
            icp.setMaxCorrespondenceDistance(0.1)
            icp.setTransformationEpsilon(1e-10)
            icp.setMaximumIterations(K)
            icp.align()
            if icp.nr_iterations_ < K:        # the alignment converged before the cap: every iteration has been seen
                full = icp.nr_iterations_
                break
            want = otree10m.correspondences(cur, 0.1)
            assert_same_correspondences(icp.fetchCorrespondences(), want, "device-driven iteration %d" % K)
            cur = orc.transform_cloud(icp.getLastIncrementalTransformation(), cur, order=1)
        assert full is not None and 3 <= full <= 5
    finally:
        gpu.setOption("icp_lookahead", 1)


# ------------------------------------------------------------------------------------------------
# the same all-iteration check OFF the bench's geometry (VERDICT r4 #1b): a volume, two close layers, 100x density
# contrast -- at sizes whose trees are as deep as the bench's, one of them with the other parity of four-way rounds
# (2.5M points: 256-point cells are strips there).  pcl_amd/synth.py: family_cloud.  PCL's own search tests use
# volumetric random clouds (test/search/test_search.cpp:292-364).
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,n,mode", [("cube", 10_000_000, 0), ("layers", 10_000_000, 1), ("clusters", 2_500_000, 0)])
def test_families_correspondences_bit_exact_at_size(gpu, orc, kind, n, mode):
    import torch
    import pcl_amd
    tgt, src, _ = pcl_amd.synth.family_pair(kind, n)
    otree = orc.KdTree(tgt)
    kw = dict(ICP_KW, max_iterations=6)
    onrm = None
    if mode == 1:
        onrm, nan = otree.normals(tgt, 8, viewpoint=(0, 0, 10))
        assert nan == 0
    ref = orc.icp_align(otree, tgt, src, mode=mode, tgt_normals=onrm, record=True, acc_double=1, **kw)
    assert ref["iterations"] >= 3
    tgt_d = torch.from_numpy(tgt).cuda()
    tree = pcl_amd.KdTree(gpu)
    tree.setInputCloud(tgt_d)
    # k = 8 on a quarter of the cloud (self-queries through the batched kernel)
    q = np.ascontiguousarray(tgt[: n // 4])
    gi, gd = tree.nearestKSearch(q, 8)
    oi, od = otree.knn(q, 8)
    assert np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
    cls = pcl_amd.IterativeClosestPointWithNormals if mode == 1 else pcl_amd.IterativeClosestPoint
    icp = cls(gpu)
    icp.setSearchMethodTarget(tree, True)
    if mode == 1:
        icp.setTargetNormals(onrm)
    icp.setInputSource(torch.from_numpy(src).cuda())
    icp.reset()
    T_prev = np.eye(4, dtype=np.float32)
    cur = src.copy()
    for it in range(ref["iterations"]):            # the cold launch and every seeded one, all n correspondences each
        sums = icp.iterate(T_prev, max_dist=0.1)
        cur = orc.transform_cloud(T_prev, cur, order=mode)
        want = otree.correspondences(cur, 0.1)
        assert_same_correspondences(icp.fetchCorrespondences(), want, "%s iteration %d" % (kind, it))
        assert int(sums[28]) == len(want[0])
        T_prev = ref["per_iter_T"][it]
    # the device-driven loop on the same clouds: same iteration count and state, the 4x4 within the contract
    icp2 = cls(gpu)
    icp2.setSearchMethodTarget(tree, True)
    if mode == 1:
        icp2.setTargetNormals(onrm)
    icp2.setInputSource(torch.from_numpy(src).cuda())
    icp2.setMaximumIterations(kw["max_iterations"])
    icp2.setMaxCorrespondenceDistance(0.1)
    icp2.setTransformationEpsilon(1e-10)
    icp2.align()
    assert icp2.nr_iterations_ == ref["iterations"]
    err = frob(icp2.getFinalTransformation(), ref["T"])
    print("%s (%d points): %d iterations, |T_gpu - T_oracle|_F = %.3g" % (kind, n, ref["iterations"], err))
    assert err < 1e-5
    if mode == 0:
        # a-5 (TransformationEstimationSVD): the device sums umeyama's moments in fp64, the reference in float.  At 2^20
        # points the 1e-5 contract also holds against the oracle's reference-ORDER float sums (config 2 above); at these
        # sizes it cannot hold against ANY double-sum implementation, this one included: the float-sum variant of the
        # oracle itself ends 5e-5 (10M cube) / 6e-5 (2.5M clusters) away from its own double-sum variant after the same six
        # iterations (measured, CPU, round 6) -- summation noise of 10^7 float terms, not the search (the correspondences
        # above are bit-equal in every iteration).  The oracle's float variant adds sequentially; Eigen's redux behind
        # umeyama adds packet-wise, which is the more accurate order, so this is the pessimistic end of what real PCL
        # would show.  Pinned here: the device stays within the contract of the double-sum oracle (above), and its
        # distance to the float-order oracle IS that variant's own noise (triangle inequality, not a new tolerance).
        ref_f = orc.icp_align(otree, tgt, src, mode=0, acc_double=0, **kw)
        gap = frob(ref_f["T"], ref["T"])
        to_float_order = frob(icp2.getFinalTransformation(), ref_f["T"])
        print("%s (%d points): float-sum vs double-sum oracle gap = %.3g ; |T_gpu - T_oracle(float order)|_F = %.3g"
              % (kind, n, gap, to_float_order))
        assert ref_f["iterations"] == ref["iterations"]
        assert abs(to_float_order - gap) < 1e-5          # the device sits on the double-sum side of that gap
        assert gap < 5e-4                                # ... which stays float noise, two orders below the pose itself
