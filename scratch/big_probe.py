"""100M-point index: deeper tree (5 box levels, level 4 not in the LDS cache); exactness spot checks."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import pcl_amd
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = pcl_amd.Context(0)
t0 = time.perf_counter()
tgt_h = synth.gaussian_surface(n, synth.TARGET_SEED)
print("gen %.1f s" % (time.perf_counter() - t0))
tgt = torch.from_numpy(tgt_h).cuda()
tree = pcl_amd.KdTree(ctx); tree.setInputCloud(tgt)
print("build ms %.1f" % tree.build_ms())
# self queries of a strided sample: index itself at distance 0
sel = torch.arange(0, n, 97, device="cuda")
q = tgt[sel].contiguous()
idx, d2 = tree.nearestKSearch(q, 1)
bad = (idx[:, 0].long() != sel) & ~((tgt[idx[:, 0].long(), :3] == q[:, :3]).all(1))
print("self-query sample %d: nonzero d2 %d, wrong idx (non-duplicate) %d" % (len(sel), int((d2 != 0).sum()), int(bad.sum())))
# brute-force check of k=4 for 256 random queries against the full cloud (torch on device)
rng = np.random.default_rng(0)
qq = torch.from_numpy(synth.gaussian_surface(256, synth.SOURCE_SEED)).cuda()
gi, gd = tree.nearestKSearch(qq, 4)
ok = True
for j in range(256):
    d = tgt[:, :3] - qq[j, :3]
    dd = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    v, i = torch.topk(dd, 4, largest=False)
    o = torch.argsort(v * 1.0, stable=True)
    if not torch.equal(v[o], gd[j]):
        ok = False; print("mismatch at", j, v[o].tolist(), gd[j].tolist()); break
print("brute-force k=4 distances equal:", ok)
src = torch.from_numpy(synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface(10_000_000, synth.SOURCE_SEED))).cuda()
icp = pcl_amd.IterativeClosestPoint(ctx); icp.setSearchMethodTarget(tree); icp.setInputSource(src); icp.reset()
T = np.eye(4, dtype=np.float32)
for it in range(4):
    sums = icp.iterate(T, max_dist=0.1); T = icp.solve(sums)
    print("it%d ms %.3f corr %d mse %.3e" % (it, icp.lastKernelMs(), sums[28], sums[27] / max(sums[28], 1)))
