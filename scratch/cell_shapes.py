"""Aspect ratios of the kd cells (aligned runs of 16 ... 4096 positions of the index order) on a uniform sheet, for cloud sizes of
both parities of the number of four-way rounds.  CPU: PCLHIP_LIB=tests/wavesim/libpclhip_wavesim.so PCLHIP_ALLOW_WAVESIM=1."""
import sys, numpy as np
sys.path.insert(0, ".")
import pcl_amd.api as A
ctx = A.Context()
rng = np.random.default_rng(1)
for n in [int(x) for x in sys.argv[1:]] or [50_000, 200_000]:
    cloud = np.ones((n, 4), np.float32)
    cloud[:, :2] = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    cloud[:, 2] = 0.5
    t = A.KdTree(ctx); t.setInputCloud(cloud)
    pts = cloud[np.asarray(t.order()), :2].astype(np.float64)
    nleaf = -(-n // 16); R = 0; cap = 1
    while cap < nleaf: cap *= 4; R += 1
    row = []
    for run in (16, 32, 64, 128, 256, 1024, 4096, 16384):
        m = n // run * run
        if m == 0: break
        c = pts[:m].reshape(-1, run, 2)
        ext = c.max(axis=1) - c.min(axis=1)
        row.append("%d: %.2f" % (run, np.median(ext.max(axis=1) / np.maximum(ext.min(axis=1), 1e-12))))
    print("n = %d (R = %d four-way levels in all): median aspect of the cells' boxes  " % (n, R) + "  ".join(row))
