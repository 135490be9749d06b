"""Dump the kd order (pclhip_index_order) and the source ordering ranks for a set of clouds; run once per library and
compare the .npz files: the index build must give the same tree bit for bit across refactors of index_build.hip.
usage: PCLHIP_LIB=... PCLHIP_ALLOW_WAVESIM=1 python scratch/order_dump.py out.npz [sizes...]"""
import sys, numpy as np
sys.path.insert(0, ".")
import pcl_amd.api as A

out = sys.argv[1]
sizes = [int(x) for x in sys.argv[2:]] or [10, 17, 100, 4096, 4097, 5000, 16385, 20000, 70000, 270000]
ctx = A.Context()
res = {}
for n in sizes:
    for kind in ("sheet", "ties", "nan"):
        rng = np.random.default_rng(n * 7 + len(kind))
        if kind == "sheet":
            xy = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
            p = np.c_[xy, (0.2 * np.sin(3 * xy[:, 0]) * np.cos(2 * xy[:, 1])).astype(np.float32)]
        elif kind == "ties":
            p = rng.integers(0, 7, (n, 3)).astype(np.float32)      # heavy ties at every splitter
        else:
            p = rng.normal(size=(n, 3)).astype(np.float32)
            p[rng.integers(0, n, max(1, n // 50))] = np.nan
        p = np.ascontiguousarray(p, dtype=np.float32)
        t = A.KdTree(ctx)
        t.setInputCloud(p)
        res["%s_%d" % (kind, n)] = np.asarray(t.order())
np.savez(out, **res)
print("wrote", out, len(res))
