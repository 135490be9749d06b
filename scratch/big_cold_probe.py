"""cold-iteration time and stand-off exit reasons at a given size: python scratch/big_cold_probe.py [n]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import pcl_amd
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = pcl_amd.Context(0)
tgt = torch.from_numpy(synth.gaussian_surface(n, synth.TARGET_SEED)).cuda()
src = torch.from_numpy(synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface(n, synth.SOURCE_SEED))).cuda()
tree = pcl_amd.KdTree(ctx); tree.setInputCloud(tgt)
ne = pcl_amd.NormalEstimation(ctx); ne.setInputCloud(tgt); ne.setSearchMethod(tree); ne.setKSearch(8); ne.setViewPoint(0, 0, 10); ne.compute(want_output=False)
icp = pcl_amd.IterativeClosestPointWithNormals(ctx); icp.setSearchMethodTarget(tree, True); icp.setInputSource(src)
ctx.stats(True)
for rep in range(2):
    icp.reset(); T = np.eye(4, dtype=np.float32)
    sums = icp.iterate(T, max_dist=0.1)
    st = ctx.stats(True); g = max(st["groups"], 1)
    print("n=%d cold ms %.2f" % (n, icp.lastKernelMs()), {k: round(v / g, 3) for k, v in st.items()})
