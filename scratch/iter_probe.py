"""per-iteration kernel times of the bench workload (no statistics counters)"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import pcl_amd
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = pcl_amd.Context(0)
tgt = torch.from_numpy(synth.gaussian_surface(n, synth.TARGET_SEED)).cuda()
src = torch.from_numpy(synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface(n, synth.SOURCE_SEED))).cuda()
tree = pcl_amd.KdTree(ctx); tree.setInputCloud(tgt)
if mode == 1:
    ne = pcl_amd.NormalEstimation(ctx); ne.setInputCloud(tgt); ne.setSearchMethod(tree); ne.setKSearch(8); ne.setViewPoint(0, 0, 10); ne.compute(want_output=False)
    print("build ms %.2f normals ms %.3f" % (tree.build_ms(), tree.lastKernelMs()))
icp = (pcl_amd.IterativeClosestPointWithNormals if mode == 1 else pcl_amd.IterativeClosestPoint)(ctx)
icp.setSearchMethodTarget(tree); icp.setInputSource(src)
for rep in range(2):
    icp.reset(); T = np.eye(4, dtype=np.float32); ms = []
    for it in range(6):
        sums = icp.iterate(T, max_dist=0.1); T = icp.solve(sums); ms.append(icp.lastKernelMs())
    print("rep%d ms/iter:" % rep, " ".join("%.3f" % m for m in ms), " mse %.3e" % (sums[27] / max(sums[28], 1)))
