"""Parity AT CONFIG 5's OWN SIZE (BASELINE.json: "100M-point synthetic cloud Morton-slab sharded across 8xMI355X"), on one
GPU: what a rank of the 8-rank job computes, and what the 100M-point single slab computes, against the oracle's exact
search of the WHOLE 100M-point target ("what's missing" #1).

  * two ranks of eight are played on this GPU (the one with the largest halo and rank 0): kd slab + halo of the
    100M-point target cut and indexed on the device, the served-group lists ON (the device-driven loop walks only the
    groups the rank serves), k = 8 normals on the slab.  The correspondences of the launch that starts the alignment and
    of two seeded launches -- every point the rank serves, index AND float distance -- must be the oracle's over the whole
    target, and the served set must be exactly the points whose current position lies in the rank's region.
  * the 100M-point target as ONE index (5.6 GB: beyond the 0.7 GB gate of the stand-off search, so the launch without
    seeds takes the path no 10M-point test reaches), a 5M-query slice: cold and seeded launches bit for bit.

The clouds are generated on the device (pcl_amd.synth.gaussian_surface_device) and downloaded once for the oracle, so both
sides see the same floats.  Reference: CorrespondenceEstimation::determineCorrespondences
(registration/include/pcl/registration/impl/correspondence_estimation.hpp:145-218).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 100_000_000
WORLD = 8
MAX_DIST = 0.1


@pytest.fixture(scope="module")
def job():
    import torch
    import pcl_amd
    from pcl_amd import synth
    from oracle import pcl_oracle as orc
    ctx = pcl_amd.Context(0)
    tgt = synth.gaussian_surface_device(N, synth.TARGET_SEED)
    src = synth.apply_rigid_device(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface_device(N, synth.SOURCE_SEED))
    torch.cuda.synchronize()
    tgt_h = tgt.cpu().numpy()
    src_h = src.cpu().numpy()
    otree = orc.KdTree(tgt_h)
    return {"ctx": ctx, "tgt": tgt, "src": src, "tgt_h": tgt_h, "src_h": src_h, "otree": otree, "orc": orc}


def assert_rank_matches_oracle(got, cur, region, otree, what):
    q, m, d = got
    p = cur[:, :3]
    inside = np.all((p >= region[:3]) & (p < region[3:]), axis=1)       # the kernel's ownership test, same floats
    served = np.nonzero(inside)[0]
    oq, om, od = otree.correspondences(np.ascontiguousarray(cur[served]), MAX_DIST)
    want_q = served[oq]
    assert len(q) == len(want_q), (what, len(q), len(want_q))
    assert np.array_equal(q, want_q), what
    bad = np.nonzero(m != om)[0]
    assert len(bad) == 0, (what, len(bad), q[bad[:5]], m[bad[:5]], om[bad[:5]])
    assert np.array_equal(d.view(np.uint32), od.view(np.uint32)), what
    return len(q)


def test_two_ranks_of_eight_at_100m_equal_the_oracle(job):
    import pcl_amd
    from pcl_amd.dist import ShardedTarget, partition_slabs, select_region
    ctx, tgt, src, orc, otree = job["ctx"], job["tgt"], job["src"], job["orc"], job["otree"]
    regions = partition_slabs(tgt, WORLD)                       # on the device (shard_dev.hip)
    halo = [len(select_region(tgt, regions[r], MAX_DIST)) - len(select_region(tgt, regions[r], 0.0)) for r in range(WORLD)]
    ranks = sorted({int(np.argmax(halo)), 0})   # rank 0 first, the largest halo last
    ctx.setOption("icp_lookahead", 0)       # no launch queued beyond the last iteration of a capped alignment
    try:
        for r in ranks:
            st = ShardedTarget(ctx, tgt, r, WORLD, MAX_DIST, k_normals=8, viewpoint=(0, 0, 10), regions=regions)
            assert st.normals_exact and st.tree.size() < N // 4
            cur = job["src_h"].copy()
            total = 0
            # the rank with the largest halo: the launch that starts the alignment and two seeded ones; rank 0: one seeded
            # launch after the cold one (the oracle's pass over a rank's 12.5M served points of a 100M-point target is
            # what this module's time goes into: four passes instead of six keep the GPU tier inside its budget)
            checked = (1, 2, 3) if r == ranks[-1] or len(ranks) == 1 else (2,)
            for K in (1, 2, 3)[:max(checked)]:
                icp = pcl_amd.IterativeClosestPointWithNormals(ctx)     # fresh: the criteria keep their memory across align() calls
                icp.setSearchMethodTarget(st.tree, True)
                icp.setInputSource(src)
                icp.setRegion(st.region)
                icp.setMaxCorrespondenceDistance(MAX_DIST)
                icp.setTransformationEpsilon(1e-10)
                icp.setMaximumIterations(K)
                icp.align()                                              # device-driven loop, served-group lists on
                assert icp.nr_iterations_ == K
                if K in checked:
                    total += assert_rank_matches_oracle(icp.fetchCorrespondences(), cur, st.region, otree,
                                                        "rank %d of %d, launch %d" % (r, WORLD, K))
                cur = orc.transform_cloud(icp.getLastIncrementalTransformation(), cur, order=1)
                del icp
            print("config 5 at size: rank %d of %d (slab + halo %d points, halo %d): %d correspondences equal "
                  "the oracle's over the whole %d-point target" % (r, WORLD, st.tree.size(), halo[r], total, N))
            assert total > N // (2 * WORLD)                              # the rank did have its share of the work
    finally:
        ctx.setOption("icp_lookahead", 1)


def test_single_slab_at_100m_cold_and_seeded_launches_equal_the_oracle(job):
    import pcl_amd
    ctx, tgt, orc, otree = job["ctx"], job["tgt"], job["orc"], job["otree"]
    tree = pcl_amd.KdTree(ctx)
    tree.setInputCloud(tgt)
    assert tree.size() == N
    nq = 5_000_000
    sub = np.ascontiguousarray(job["src_h"][:nq])
    icp = pcl_amd.IterativeClosestPoint(ctx)
    icp.setSearchMethodTarget(tree, True)
    icp.setInputSource(sub)
    icp.reset()
    cur = sub.copy()
    T = np.eye(4, dtype=np.float32)
    for it in range(3):                                  # the launch without seeds (the gated path), then two seeded ones
        sums = icp.iterate(T, max_dist=MAX_DIST)
        cur = orc.transform_cloud(T, cur, order=0)
        oq, om, od = otree.correspondences(cur, MAX_DIST)
        q, m, d = icp.fetchCorrespondences()
        assert np.array_equal(q, oq) and np.array_equal(m, om), (it, int((m != om).sum()) if len(m) == len(om) else -1)
        assert np.array_equal(d.view(np.uint32), od.view(np.uint32)), it
        T = icp.solve(sums)
    print("config 5 at size: the %d-point single slab, %d queries x 3 launches equal the oracle's" % (N, nq))
