"""A/B of the index build variants: same tree?  (work counters of the normals and ICP searches + timings)"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import pcl_amd
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = pcl_amd.Context(0)
tgt = torch.from_numpy(synth.gaussian_surface(n, synth.TARGET_SEED)).cuda()
for rep in range(3):
    tree = pcl_amd.KdTree(ctx); tree.setInputCloud(tgt)
    ctx.stats(True)
    ne = pcl_amd.NormalEstimation(ctx); ne.setInputCloud(tgt); ne.setSearchMethod(tree); ne.setKSearch(8); ne.setViewPoint(0, 0, 10)
    ne.compute(want_output=False)
    st = ctx.stats(True)
    ctx.stats(False)
    ne.compute(want_output=False)
    q = tgt[::1000].cpu().numpy().copy()
    idx, d2 = tree.nearestKSearch(q, 8)
    print("rep %d build %.3f ms normals %.3f ms (with counters) / %.3f ms" % (rep, tree.build_ms(), 0.0, tree.lastKernelMs()),
          {k: int(v) for k, v in st.items()}, "knn checksum", int(np.asarray(idx, np.int64).sum()), float(np.asarray(d2, np.float64).sum()))
