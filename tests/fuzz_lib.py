"""Randomised rounds of the device against the oracle, shared by the bounded `-m gpu` slices (tests/test_gpu_fuzz.py) and the
open-ended runs (scratch/fuzz_knn.py, scratch/fuzz_filters.py).  Every function draws one random case from `rng`, runs it
through the C ABI and through the oracle, and returns (ok, one-line description)."""
import numpy as np


def _cloud(rng, n, kind):
    import pcl_amd
    if kind == 0:   p = rng.uniform(-1, 1, (n, 3))
    elif kind == 1: p = np.c_[rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), 1e-4 * rng.normal(size=n)]        # plane
    elif kind == 2: t = rng.uniform(0, 1, n); p = np.c_[t, 2 * t, -t] + 1e-5 * rng.normal(size=(n, 3))         # line
    elif kind == 3: c = rng.uniform(-1, 1, (8, 3)); p = c[rng.integers(0, 8, n)] + 1e-3 * rng.normal(size=(n, 3))  # clusters
    elif kind == 4: p = np.round(rng.uniform(0, 6, (n, 3)))                                                    # lattice: ties, duplicates
    elif kind == 5: p = rng.uniform(-1, 1, (n, 3)) * np.array([100.0, 1.0, 0.01])                              # anisotropic
    else:           p = pcl_amd.synth.gaussian_surface(n, int(rng.integers(1, 1 << 30)))[:, :3]
    p = p * float(10 ** rng.uniform(-3, 3)) + rng.uniform(-1, 1, 3) * float(10 ** rng.uniform(-2, 4)) * (rng.random() < 0.5)
    out = np.ones((n, 4), np.float32)
    out[:, :3] = p.astype(np.float32)
    if n > 10 and rng.random() < 0.3:
        out[rng.integers(0, n, max(1, n // 50)), rng.integers(0, 3)] = np.nan
    return out


def knn_icp_round(ctx, orc, rng, sizes=(1, 2, 15, 16, 17, 63, 64, 65, 1000, 4096, 4097, 20000, 70000, 250000)):
    """k-NN lists, then the unseeded and the seeded ICP correspondences of one random target / query pair."""
    import pcl_amd
    n = int(rng.choice(sizes))
    kind = int(rng.integers(0, 7))
    tgt = _cloud(rng, n, kind)
    nq = int(rng.choice([1, 63, 64, 65, 5000, 40000]))
    qk = int(rng.integers(0, 3))
    if qk == 0:   qry = _cloud(rng, nq, kind)
    elif qk == 1: qry = tgt[rng.integers(0, n, nq)].copy()
    else:
        qry = tgt[rng.integers(0, n, nq)].copy()
        ext = np.nanmax(np.abs(tgt[:, :3] - np.nanmean(tgt[:, :3], 0)), 0) + 1e-6
        qry[:, :3] += (rng.normal(size=(nq, 3)) * ext * float(10 ** rng.uniform(-3, 1))).astype(np.float32)
    k = int(rng.choice([1, 2, 5, 8, 13, 16, 31, 32, 33, 64]))
    tree = pcl_amd.KdTree(ctx)
    tree.setInputCloud(tgt)
    otree = orc.KdTree(tgt)
    gi, gd = tree.nearestKSearch(qry, k)
    oi, od = otree.knn(qry, k)
    ok = np.array_equal(gi, oi) and np.array_equal(gd, od, equal_nan=True)
    msg = "n=%6d kind=%d nq=%5d q=%d k=%2d knn %s" % (n, kind, nq, qk, k, "ok" if ok else "MISMATCH")
    # ICP correspondences: unseeded, then seeded by the first pass after a small motion
    fin = np.isfinite(tgt[:, :3]).all(1).sum()
    if fin >= 1:
        icp = pcl_amd.IterativeClosestPoint(ctx)
        icp.setSearchMethodTarget(tree, True)
        icp.setInputSource(qry)
        icp.reset()
        I = np.eye(4, dtype=np.float32)
        scale = float(np.nanmax(np.abs(tgt[:, :3])) + 1e-6)
        md = float(10 ** rng.uniform(-2, 1)) * scale if rng.random() < 0.7 else None
        big = np.sqrt(np.finfo(np.float64).max)
        icp.iterate(I, max_dist=md)
        q1, m1, d1 = icp.fetchCorrespondences()
        oq, om, od1 = otree.correspondences(qry, max_dist=md if md is not None else big)
        ok1 = np.array_equal(q1, oq) and np.array_equal(m1, om) and np.array_equal(d1, od1)
        T = np.eye(4, dtype=np.float32)
        T[:3, 3] = (rng.normal(size=3) * 1e-3 * scale).astype(np.float32)
        icp.iterate(T, max_dist=md)
        q2, m2, d2 = icp.fetchCorrespondences()
        moved = icp.transformCloud(qry, T)
        oq2, om2, od2 = otree.correspondences(moved, max_dist=md if md is not None else big)
        ok2 = np.array_equal(q2, oq2) and np.array_equal(m2, om2) and np.array_equal(d2, od2)
        # three more iterations with shrinking motions, as a converging alignment makes them (builds that keep the groups'
        # leaf lists across iterations -- traverse.hpp: GroupRec -- search from their records here)
        ok3 = True
        for step in (1e-4, 1e-5, 0.0):
            T = np.eye(4, dtype=np.float32)
            T[:3, 3] = (rng.normal(size=3) * step * scale).astype(np.float32)
            icp.iterate(T, max_dist=md)
            q3, m3, d3 = icp.fetchCorrespondences()
            moved = icp.transformCloud(moved, T)
            oq3, om3, od3 = otree.correspondences(moved, max_dist=md if md is not None else big)
            ok3 = ok3 and np.array_equal(q3, oq3) and np.array_equal(m3, om3) and np.array_equal(d3, od3)
        msg += "  icp cold %s seeded %s converging %s" % ("ok" if ok1 else "MISMATCH", "ok" if ok2 else "MISMATCH",
                                                        "ok" if ok3 else "MISMATCH")
        ok = ok and ok1 and ok2 and ok3
    return ok, msg


FIELDS = {"x": 0, "y": 1, "z": 2, "normal_x": 4, "normal_y": 5, "normal_z": 6, "curvature": 8}


def filters_round(ctx, orc, rng, sizes=(1, 50, 3000, 40000, 300000)):
    """VoxelGrid (leaf sizes, field filter, negative limits, minimum points, both record layouts) and NormalEstimation with
    a search surface / index subset of one random cloud."""
    import pcl_amd
    n = int(rng.choice(sizes))
    wide = rng.random() < 0.5
    cloud = np.zeros((n, 12 if wide else 4), np.float32)
    kind = int(rng.integers(0, 3))
    if kind == 0:   xyz = rng.uniform(-1, 1, (n, 3))
    elif kind == 1: xyz = pcl_amd.synth.gaussian_surface(n, int(rng.integers(1, 1 << 30)))[:, :3]
    else:           xyz = rng.normal(size=(n, 3)) * np.array([5.0, 0.2, 1.0])
    cloud[:, :3] = (xyz * float(10 ** rng.uniform(-1, 1)) + rng.uniform(-3, 3, 3)).astype(np.float32)
    cloud[:, 3] = 1
    if wide:
        cloud[:, 4:7] = rng.normal(size=(n, 3)).astype(np.float32)
        cloud[:, 8] = rng.uniform(0, 1, n).astype(np.float32)
    if n > 20 and rng.random() < 0.3:
        cloud[rng.integers(0, n, 3), rng.integers(0, 3)] = np.nan
    ext = float(np.nanmax(cloud[:, :3]) - np.nanmin(cloud[:, :3])) + 1e-6
    leaf = (ext * 10 ** rng.uniform(-2.2, -0.3, 3)).astype(np.float32)
    vg = pcl_amd.VoxelGrid(ctx)
    vg.setInputCloud(cloud)
    vg.setLeafSize(*[float(v) for v in leaf])
    minpts = int(rng.choice([0, 0, 1, 2, 5]))
    vg.setMinimumPointsNumberPerVoxel(minpts)
    kw = {}
    if rng.random() < 0.6:
        name = str(rng.choice([f for f, c in FIELDS.items() if c < cloud.shape[1]]))
        col = FIELDS[name]
        lo, hi = sorted(np.nanquantile(cloud[:, col], rng.uniform(0, 1, 2)).tolist())
        neg = bool(rng.random() < 0.4)
        vg.setFilterFieldName(name); vg.setFilterLimits(lo, hi); vg.setFilterLimitsNegative(neg)
        kw = dict(limits=(lo, hi), field=col, negative=neg)
    vg.setDownsampleAllData(bool(rng.random() < 0.5))
    try:
        out = vg.filter()
        want, _ = orc.voxelgrid(cloud, leaf, min_points_per_voxel=minpts, **kw)
        if want is None:
            ok = False; why = "oracle refused, device did not"
        else:
            ok = np.array_equal(out[:, :4], want, equal_nan=True); why = "%d voxels" % len(want)
    except pcl_amd.PclHipError as e:
        want, _ = orc.voxelgrid(cloud, leaf, min_points_per_voxel=minpts, **kw)
        ok = want is None; why = "both refuse (overflow)" if ok else "device refused: %s" % e
    msg = "n=%6d wide=%d %s  voxelgrid %s (%s)" % (n, wide, kw, "ok" if ok else "MISMATCH", why)
    # normals at other points / an index subset, k-NN mode
    k = int(rng.choice([3, 8, 10, 20, 33]))
    if n >= 50:
        nq = int(rng.choice([1, 64, 700, 20000]))
        q = cloud[rng.integers(0, n, nq), :4].copy()
        q[:, :3] += (rng.normal(size=(nq, 3)) * ext * 1e-3).astype(np.float32)
        ind = rng.integers(0, nq, max(1, nq // 3)).astype(np.int32) if rng.random() < 0.5 else None
        ne = pcl_amd.NormalEstimation(ctx)
        ne.setInputCloud(q); ne.setSearchSurface(cloud[:, :4].copy()); ne.setKSearch(k); ne.setViewPoint(0.5, -1.0, 20.0)
        ne.setIndices(ind)
        got = ne.compute()
        surf = ne.getSearchSurface()
        want, nan = orc.KdTree(surf).normals_at(surf, q, k, viewpoint=(0.5, -1.0, 20.0), indices=ind)
        good = ~np.isnan(want[:, 0])
        okn = np.array_equal(np.isnan(got[:, 0]), ~good) and ne.nan_count == nan
        if okn and good.any():
            # ill-conditioned plane fits (near-degenerate neighbourhoods) amplify the libm differences: compare where the
            # oracle's own curvature says the plane is defined
            sel = good & (want[:, 3] < 0.2)
            dots = np.sum(got[sel, :3] * want[sel, :3], axis=1)
            okn = (not sel.any()) or (np.abs(dots).min() > 1 - 1e-3 and np.median(np.abs(got[sel, 3] - want[sel, 3])) < 1e-5)
        msg += "  normals_at k=%d nq=%d idx=%s %s" % (k, nq, ind is not None, "ok" if okn else "MISMATCH")
        ok = ok and okn
    return ok, msg
