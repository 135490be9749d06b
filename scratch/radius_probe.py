"""radiusSearch / radius-mode normals timings at n points (PCLHIP_LIB picks the build)."""
import os, sys, time, torch
sys.path.insert(0, '.')
import pcl_amd
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = pcl_amd.Context(0)
cloud = torch.from_numpy(synth.gaussian_surface(n, synth.TARGET_SEED)).cuda()
tree = pcl_amd.KdTree(ctx)
tree.setInputCloud(cloud)
print(os.environ.get("PCLHIP_LIB", "libpclhip.so"))
q = cloud[:1_000_000]
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); tree.radiusSearch(q, 0.002); torch.cuda.synchronize()
    print("radiusSearch 1M queries r=0.002: %.2f ms wall" % ((time.perf_counter() - t0) * 1e3))
ne = pcl_amd.NormalEstimation(ctx); ne.setInputCloud(cloud); ne.setSearchMethod(tree); ne.setRadiusSearch(0.002); ne.setViewPoint(0, 0, 10)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ne.compute(want_output=False); torch.cuda.synchronize()
    print("normals radius 0.002 (10M): %.2f ms wall, kernels %.2f ms" % ((time.perf_counter() - t0) * 1e3, tree.lastKernelMs()))
