import sys, numpy as np, torch
sys.path.insert(0, '.')
import pcl_amd
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ctx = pcl_amd.Context(0)
tgt = torch.from_numpy(synth.gaussian_surface(n, synth.TARGET_SEED)).cuda()
src = torch.from_numpy(synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface(n, synth.SOURCE_SEED))).cuda()
tree = pcl_amd.KdTree(ctx); tree.setInputCloud(tgt)
print("build ms", tree.build_ms())
ctx.stats(True)
ne = pcl_amd.NormalEstimation(ctx); ne.setInputCloud(tgt); ne.setSearchMethod(tree); ne.setKSearch(8); ne.setViewPoint(0,0,10); ne.compute(want_output=False)
st = ctx.stats(True); g = max(st["groups"],1)
print("normals k=8: ms %.3f" % tree.lastKernelMs(), {k: round(v/g,2) for k,v in st.items()})
icp = pcl_amd.IterativeClosestPointWithNormals(ctx); icp.setSearchMethodTarget(tree); icp.setInputSource(src)
icp.reset(); T = np.eye(4, dtype=np.float32)
for it in range(5):
    sums = icp.iterate(T, max_dist=0.1); T = icp.solve(sums)
    st = ctx.stats(True); g = max(st["groups"],1)
    print("icp it%d: kernel ms %.3f corr %d mse %.3e" % (it, icp.lastKernelMs(), sums[28], sums[27]/max(sums[28],1)), {k: round(v/g,2) for k,v in st.items()})
