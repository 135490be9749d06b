"""clock64 ticks per stage of the seeded search body per 64-query group and launch (-DPCLHIP_TRAV_PROFILE variant)"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import pcl_amd
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = pcl_amd.Context(0)
tgt = torch.from_numpy(synth.gaussian_surface(n, synth.TARGET_SEED)).cuda()
src = torch.from_numpy(synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface(n, synth.SOURCE_SEED))).cuda()
tree = pcl_amd.KdTree(ctx); tree.setInputCloud(tgt)
ne = pcl_amd.NormalEstimation(ctx); ne.setInputCloud(tgt); ne.setSearchMethod(tree); ne.setKSearch(8); ne.setViewPoint(0, 0, 10); ne.compute(want_output=False)
icp = pcl_amd.IterativeClosestPointWithNormals(ctx)
icp.setSearchMethodTarget(tree); icp.setInputSource(src)
icp.reset(); T = np.eye(4, dtype=np.float32)
groups = (n + 63) // 64
names = ["prologue", "box+start", "node scans", "lists/pushes", "box tests", "rounds", "resolve", "index+ties+stores+next"]
ctx.counters(True)
for it in range(7):
    sums = icp.iterate(T, max_dist=0.1); T = icp.solve(sums)
    c = ctx.counters(True)
    tot = sum(c)
    print("launch %d: %.3f ms, ticks per group %6.0f | " % (it + 1, icp.lastKernelMs(), tot / groups) +
          "  ".join("%s %.0f (%.0f%%)" % (nm, v / groups, 100.0 * v / max(tot, 1)) for nm, v in zip(names, c)))
