"""The per-lane seeded search (pcl_amd/csrc/lane.hip, lane_search.hpp) against the oracle, and the structure it rests on.

Two things are checked here, on geometries the bench does not have (volumes, two close layers, 100x density contrast,
lattices full of ties, tiny clouds):

  * the CELLS themselves (pclhip_index_cells): the claim the search ends on -- "no point of any other node lies in the
    interior of a node's cell" -- is verified point by point, level by level; and
  * the correspondences of seeded launches (index and float distance bits) for every setting of the two knobs that decide
    which code a query runs through: how many quad levels the first pass climbs (0: everything that does not fit its own
    leaf's cell goes to the finishing pass; 15: nothing does) and from which distance a seed is replaced by a greedy
    descent (0: always; huge: never).

Reference behaviour: CorrespondenceEstimation::determineCorrespondences
(registration/include/pcl/registration/impl/correspondence_estimation.hpp:145-218) over KdTreeFLANN::nearestKSearch.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import pcl_amd
    return pcl_amd.Context(0)


@pytest.fixture
def lane(gpu):
    """the per-lane search is an OPTION of the context (off by default since its first measurement: exact, but 2.5x slower
    than the wave-cooperative body at 10M points -- profiles/r05_lane_search_ab.txt); these tests switch it on"""
    gpu.setOption("lane_search", 1)
    yield gpu
    gpu.setOption("lane_search", 0)
    gpu.setOption("lane_max_up", 2)
    gpu.setOption("lane_far", 0.25)


@pytest.fixture(scope="module")
def orc():
    from oracle import pcl_oracle
    return pcl_oracle


def make_cloud(kind, n, seed):
    from pcl_amd import synth
    rng = np.random.default_rng(seed)
    if kind in synth.FAMILIES:
        return synth.family_cloud(kind, n, seed)
    out = np.ones((n, 4), np.float32)
    if kind == "lattice":                       # heavy ties at every cut and at every distance
        out[:, :3] = rng.integers(0, 9, (n, 3)).astype(np.float32)
    elif kind == "plane":                       # one axis without extent
        out[:, :2] = rng.uniform(0, 1, (n, 2)).astype(np.float32)
        out[:, 2] = 0.5
    elif kind == "line":
        out[:, 0] = rng.uniform(0, 1, n).astype(np.float32)
        out[:, 1:3] = 0.25
    elif kind == "far":                         # a unit scene 1e4 from the origin: float spacing 1e-3
        out[:, :3] = (rng.uniform(0, 1, (n, 3)) + 1e4).astype(np.float32)
    else:
        raise ValueError(kind)
    return out


def rigid(rx=0.0, ry=0.0, rz=0.0, t=(0, 0, 0)):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = t
    return T.astype(np.float32)


CELL_CASES = [("sheet", 70_001), ("cube", 50_000), ("layers", 40_000), ("clusters", 60_000), ("lattice", 20_000),
              ("plane", 30_000), ("line", 5_000), ("far", 20_000), ("cube", 1), ("cube", 15), ("cube", 17), ("cube", 64),
              ("cube", 65), ("sheet", 1025), ("sheet", 4097), ("sheet", 16_385),
              # more than 4096 nodes on the first quad level: the large levels by their own launches around the fused
              # kernel of the small ones (index_build.hip: quad_top_kernel)
              ("sheet", 300_001), ("cube", 1_200_000)]


@pytest.mark.parametrize("kind,n", CELL_CASES)
def test_cells_hold_no_foreign_point(gpu, kind, n):
    import pcl_amd
    cloud = make_cloud(kind, n, 7)
    tree = pcl_amd.KdTree(gpu)
    tree.setInputCloud(cloud)
    order = tree.order()
    pts = cloud[order, :3]
    nleaf = (n + 15) // 16
    _, _, top = tree.cells(0)
    rng = np.random.default_rng(n)
    q = 0
    while True:
        boxes, cells, top2 = tree.cells(q)
        assert top2 == top
        cnt = (nleaf + 4 ** q - 1) // 4 ** q
        assert len(cells) == cnt
        span = 16 * 4 ** q
        npick = 250 if n <= 100_000 else 30     # (every pick compares the whole cloud against the node's cell)
        pick = np.arange(cnt) if cnt <= npick else np.unique(np.concatenate([rng.integers(0, cnt, npick), [0, cnt - 1, cnt // 2]]))
        for i in pick:
            a, b = i * span, min(n, (i + 1) * span)
            inside = pts[a:b]
            # the node's box is the tight box of its points, and lies in the (closed) cell
            assert np.array_equal(boxes[i, :3], inside.min(axis=0)) and np.array_equal(boxes[i, 3:], inside.max(axis=0)), (q, i)
            lo, hi = cells[i, :3], cells[i, 3:]
            assert np.all(lo <= hi), ("inverted cell", q, i)          # the build separated every pair of siblings
            assert np.all(inside >= lo) and np.all(inside <= hi), (q, i)
            # no foreign point in the interior
            outside = np.concatenate([pts[:a], pts[b:]])
            if len(outside):
                in_interior = np.all((outside > lo) & (outside < hi), axis=1)
                assert not in_interior.any(), (q, i, int(in_interior.sum()))
        if cnt == 1:
            assert q == top
            assert np.all(np.isinf(cells[0]))                         # the root's cell is all of space
            break
        q += 1


def run_iterations(gpu, orc, tgt, src, steps, max_dist, mode=0):
    """host-driven iterations (the first one is the launch that starts an alignment, the others are seeded) against the
    oracle's exact correspondences of the same moved cloud: queries, matches and float distances bit for bit"""
    import pcl_amd
    otree = orc.KdTree(tgt)
    icp = pcl_amd.IterativeClosestPoint(gpu)
    icp.setInputTarget(tgt)
    icp.setInputSource(src)
    icp.reset()
    cur = src.copy()
    for it, T in enumerate(steps):
        icp.iterate(T, max_dist=max_dist)
        q, m, d = icp.fetchCorrespondences()
        cur = orc.transform_cloud(T, cur, order=0)
        oq, om, od = otree.correspondences(cur, max_dist)
        assert np.array_equal(q, oq), (it, len(q), len(oq))
        assert np.array_equal(m, om), (it, int((m != om).sum()))
        assert np.array_equal(d, od), it


STEPS = [np.eye(4, dtype=np.float32),                       # cold launch
         rigid(t=(1e-4, -2e-4, 1e-4)),                      # a nudge: seeds next to the answers
         rigid(rz=0.03, t=(0.02, 0.01, -0.01)),             # a slide of tens of spacings: seeds are far
         np.eye(4, dtype=np.float32),                       # nothing moves: every seed IS the answer
         rigid(rx=0.5, ry=-0.3, t=(0.3, 0.2, 0.1)),         # off the target altogether
         rigid(t=(-0.3, -0.2, -0.1))]


@pytest.mark.parametrize("kind,n", [("sheet", 30_000), ("cube", 20_000), ("layers", 30_000), ("clusters", 30_000), ("far", 8_000)])
@pytest.mark.parametrize("max_up,far", [(2, 0.25), (0, 0.25), (15, 0.25), (1, 0.0), (2, 1e9)])
def test_seeded_launches_equal_the_oracle(lane, orc, kind, n, max_up, far):
    gpu = lane
    tgt = make_cloud(kind, n, 1001)
    src = make_cloud(kind, n // 2 + 3, 2002)
    src[::97, 0] = np.nan                                     # non-finite queries travel unchanged and match nothing
    gpu.setOption("lane_max_up", max_up)
    gpu.setOption("lane_far", far)
    try:
        for max_dist in (0.05, 1e3):
            run_iterations(gpu, orc, tgt, src, STEPS, max_dist)
    finally:
        gpu.setOption("lane_max_up", 2)
        gpu.setOption("lane_far", 0.25)


@pytest.mark.parametrize("max_up,far", [(2, 0.25), (0, 0.0), (15, 1e9)])
def test_seeded_launches_on_a_lattice_full_of_ties(lane, orc, max_up, far):
    gpu = lane
    # every query between lattice points has 2, 4 or 8 equidistant targets; the index order is unrelated to the position,
    # so "lowest index" is not "first visited".  Ties are the finishing pass's business: all of them have to get there.
    g = np.arange(14, dtype=np.float32)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    tgt = np.stack([X.ravel(), Y.ravel(), Z.ravel(), np.ones(X.size, np.float32)], 1)
    tgt = np.ascontiguousarray(tgt[np.random.default_rng(3).permutation(len(tgt))])
    src = tgt[::2].copy()
    steps = [rigid(t=(0.5, 0, 0)), rigid(t=(0, 0.5, 0)), rigid(t=(0, 0, 0.5)), rigid(t=(0.25, 0, 0)), np.eye(4, dtype=np.float32),
             rigid(t=(-0.75, -0.5, -0.5))]
    gpu.setOption("lane_max_up", max_up)
    gpu.setOption("lane_far", far)
    try:
        run_iterations(gpu, orc, tgt, src, steps, 10.0)
    finally:
        gpu.setOption("lane_max_up", 2)
        gpu.setOption("lane_far", 0.25)


@pytest.mark.parametrize("n_tgt", [1, 2, 15, 16, 17, 63, 65, 300])
def test_seeded_launches_on_tiny_targets(lane, orc, n_tgt):
    gpu = lane
    rng = np.random.default_rng(n_tgt)
    tgt = np.ones((n_tgt, 4), np.float32)
    tgt[:, :3] = rng.normal(size=(n_tgt, 3)).astype(np.float32)
    src = np.ones((500, 4), np.float32)
    src[:, :3] = rng.normal(size=(500, 3)).astype(np.float32)
    for max_dist in (0.3, 1e3):
        run_iterations(gpu, orc, tgt, src, STEPS[:4], max_dist)


def test_lane_search_equals_the_wave_cooperative_search_in_the_device_loop(gpu):
    # the device-driven loop end to end with the option off (the wave-cooperative body of search.hip: round 4's path) and on:
    # same iteration count, same state, same 4x4 bit for bit (the matches are the same, the sums run in the same order)
    import pcl_amd
    tgt, src, _ = pcl_amd.synth.family_pair("layers", 40_000)
    res = []
    for on in (0, 1):
        gpu.setOption("lane_search", on)
        try:
            icp = pcl_amd.IterativeClosestPoint(gpu)
            icp.setInputTarget(tgt)
            icp.setInputSource(src)
            icp.setMaximumIterations(12)
            icp.setMaxCorrespondenceDistance(0.1)
            icp.setTransformationEpsilon(1e-10)
            icp.align()
            res.append((icp.nr_iterations_, icp.getFinalTransformation().copy(), icp.hasConverged()))
        finally:
            gpu.setOption("lane_search", 0)
    assert res[0][0] == res[1][0] and res[0][2] == res[1][2]
    assert np.array_equal(res[0][1], res[1][1])


def test_lane_search_counters(lane):
    gpu = lane
    # pclhip_ctx_stats with the lane kernels: slot 0 = queries, 2 = done in the first pass, 4 = finished by the second;
    # on a converged sheet most queries are done by looking at one leaf (what the design rests on)
    import pcl_amd
    tgt, src, T = pcl_amd.synth.icp_pair(60_000)
    icp = pcl_amd.IterativeClosestPoint(gpu)
    icp.setInputTarget(tgt)
    icp.setInputSource(pcl_amd.synth.apply_rigid(T, src))        # aligned already
    icp.reset()
    icp.iterate(np.eye(4, dtype=np.float32), max_dist=0.1)
    gpu.counters(True)
    icp.iterate(np.eye(4, dtype=np.float32), max_dist=0.1)
    c = gpu.counters(False)
    assert c[0] == 60_000
    assert c[2] + c[4] == c[0]                                    # every query is finished by exactly one of the passes
    assert c[1] > 0.4 * c[0] and c[2] > 0.7 * c[0], list(c)
    assert c[3] == 0                                              # no seed was far


def test_index_cells_argument_checks(gpu):
    import pcl_amd
    cloud = make_cloud("cube", 5000, 3)
    tree = pcl_amd.KdTree(gpu)
    tree.setInputCloud(cloud)
    boxes, cells, top = tree.cells(0)
    counts = [len(tree.cells(q)[0]) for q in range(top + 1)]
    assert top == 5 and counts == [313, 79, 20, 5, 2, 1]           # ceil(313 / 4^q): the levels lie one after the other
    with pytest.raises(pcl_amd.PclHipError, match="level out of range"):
        tree.cells(top + 1)
    with pytest.raises(pcl_amd.PclHipError, match="level out of range"):
        tree.cells(-1)
