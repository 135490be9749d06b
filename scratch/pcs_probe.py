"""workload for PC sampling of the seeded search kernel: an alignment, then REPS more (converged) iterations, host-driven"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import pcl_amd
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ctx = pcl_amd.Context(0)
tgt = torch.from_numpy(synth.gaussian_surface(n, synth.TARGET_SEED)).cuda()
src = torch.from_numpy(synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface(n, synth.SOURCE_SEED))).cuda()
tree = pcl_amd.KdTree(ctx); tree.setInputCloud(tgt)
ne = pcl_amd.NormalEstimation(ctx); ne.setInputCloud(tgt); ne.setSearchMethod(tree); ne.setKSearch(8); ne.setViewPoint(0, 0, 10); ne.compute(want_output=False)
icp = pcl_amd.IterativeClosestPointWithNormals(ctx)
icp.setSearchMethodTarget(tree); icp.setInputSource(src)
icp.reset(); T = np.eye(4, dtype=np.float32); ms = []
for it in range(5 + reps):
    sums = icp.iterate(T, max_dist=0.1); T = icp.solve(sums); ms.append(icp.lastKernelMs())
print("ms/iter:", " ".join("%.3f" % m for m in ms[:8]), "... converged mean %.4f" % (sum(ms[5:]) / max(len(ms) - 5, 1)))
