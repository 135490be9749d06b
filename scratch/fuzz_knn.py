"""Randomised self-check (not part of the test suite): k-NN, radius search, unseeded and seeded ICP correspondences of
random clouds of many shapes against the oracle, bit for bit.  python scratch/fuzz_knn.py [seed] [rounds]"""
import sys, numpy as np
sys.path.insert(0, '.')
import pcl_amd
from oracle import pcl_oracle as orc

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
ctx = pcl_amd.Context(0)


def cloud(n, kind):
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


bad = 0
for it in range(rounds):
    n = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 1000, 4096, 4097, 20000, 70000, 250000]))
    kind = int(rng.integers(0, 7))
    tgt = cloud(n, kind)
    nq = int(rng.choice([1, 63, 64, 65, 5000, 40000]))
    qk = int(rng.integers(0, 3))
    if qk == 0:   qry = cloud(nq, kind)
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
    msg = "it %2d n=%6d kind=%d nq=%5d q=%d k=%2d knn %s" % (it, n, kind, nq, qk, k, "ok" if ok else "MISMATCH")
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
        icp.iterate(I, max_dist=md)
        q1, m1, d1 = icp.fetchCorrespondences()
        oq, om, od1 = otree.correspondences(qry, max_dist=md if md is not None else np.sqrt(np.finfo(np.float64).max))
        ok1 = np.array_equal(q1, oq) and np.array_equal(m1, om) and np.array_equal(d1, od1)
        T = np.eye(4, dtype=np.float32)
        T[:3, 3] = (rng.normal(size=3) * 1e-3 * scale).astype(np.float32)
        icp.iterate(T, max_dist=md)
        q2, m2, d2 = icp.fetchCorrespondences()
        moved = icp.transformCloud(qry, T)
        oq2, om2, od2 = otree.correspondences(moved, max_dist=md if md is not None else np.sqrt(np.finfo(np.float64).max))
        ok2 = np.array_equal(q2, oq2) and np.array_equal(m2, om2) and np.array_equal(d2, od2)
        msg += "  icp cold %s seeded %s" % ("ok" if ok1 else "MISMATCH", "ok" if ok2 else "MISMATCH")
        ok = ok and ok1 and ok2
    bad += 0 if ok else 1
    print(msg, flush=True)
print("FUZZ seed %d: %d / %d rounds with a mismatch" % (seed, bad, rounds))
sys.exit(1 if bad else 0)
