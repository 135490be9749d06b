import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import pcl_amd
from pcl_amd import synth
n = 10_000_000
tgt = torch.from_numpy(synth.gaussian_surface(n, synth.TARGET_SEED)).cuda()
src = torch.from_numpy(synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface(n, synth.SOURCE_SEED))).cuda()
torch.cuda.synchronize()
t0 = time.perf_counter(); ctx = pcl_amd.Context(0); t1 = time.perf_counter()
tree = pcl_amd.KdTree(ctx)
tw0 = time.perf_counter(); tree.setInputCloud(tgt); ctx.synchronize(); tw1 = time.perf_counter()
first = tree.build_ms()
tree.setInputCloud(tgt); second = tree.build_ms()
ne = pcl_amd.NormalEstimation(ctx); ne.setInputCloud(tgt); ne.setSearchMethod(tree); ne.setKSearch(8); ne.setViewPoint(0,0,10)
ne.compute(want_output=False); n1 = tree.lastKernelMs(); ne.compute(want_output=False); n2 = tree.lastKernelMs()
icp = pcl_amd.IterativeClosestPointWithNormals(ctx); icp.setSearchMethodTarget(tree, True)
ts0 = time.perf_counter(); icp.setInputSource(src); ctx.synchronize(); ts1 = time.perf_counter()
print("ctx create %.1f ms | first build %.2f ms (wall %.1f) steady %.2f | normals first %.2f steady %.2f | set source wall %.1f ms order %.2f" %
      ((t1-t0)*1e3, first, (tw1-tw0)*1e3, second, n1, n2, (ts1-ts0)*1e3, icp.sourceOrderMs()))
