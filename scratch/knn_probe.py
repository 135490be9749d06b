import sys, numpy as np, torch
sys.path.insert(0, '.')
import pcl_amd
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = pcl_amd.Context(0)
tgt = torch.from_numpy(synth.gaussian_surface(n, synth.TARGET_SEED)).cuda()
tree = pcl_amd.KdTree(ctx); tree.setInputCloud(tgt)
idx = torch.empty((n, 16), dtype=torch.int32, device="cuda"); d2 = torch.empty((n, 16), dtype=torch.float32, device="cuda")
for k in (1, 8, 16):
    for rep in range(2):
        ctx.stats(True)
        tree.nearestKSearch(tgt, k)
        st = ctx.stats(True); g = max(st["groups"], 1)
    print("knn k=%d self-query: kernel ms %.3f" % (k, tree.lastKernelMs()), {a: round(v / g, 2) for a, v in st.items()})
ne = pcl_amd.NormalEstimation(ctx); ne.setInputCloud(tgt); ne.setSearchMethod(tree); ne.setKSearch(8); ne.setViewPoint(0, 0, 10)
for rep in range(2):
    ctx.stats(True); ne.compute(want_output=False); st = ctx.stats(True); g = max(st["groups"], 1)
print("normals k=8: kernel ms %.3f" % tree.lastKernelMs(), {a: round(v / g, 2) for a, v in st.items()})
