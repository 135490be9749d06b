import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import pcl_amd
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = pcl_amd.Context(0)
cloud_h = synth.gaussian_surface(n, synth.TARGET_SEED)
cloud = torch.from_numpy(cloud_h).cuda()
def timed(name, fn, reps=2):
    for r in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-40s %8.2f ms" % (name, dt * 1e3)); return out
vg = pcl_amd.VoxelGrid(ctx); vg.setInputCloud(cloud); vg.setLeafSize(0.005)
out = timed("VoxelGrid 10M leaf 0.005 (device in/out)", lambda: vg.filter())
print("   voxels:", len(out))
tree = pcl_amd.KdTree(ctx)
timed("index build (wall, incl. alloc)", lambda: (tree.__setattr__('_cloud_id', None), tree.setInputCloud(cloud)))
print("   build_ms (gpu events): %.2f" % tree.build_ms())
src = torch.from_numpy(synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface(n, synth.SOURCE_SEED))).cuda()
for kk in (1, 8):
    timed("nearestKSearch k=%d, %d queries (self)" % (kk, n), lambda: tree.nearestKSearch(cloud, kk))
    print("   kernel ms %.2f" % tree.lastKernelMs())
    timed("nearestKSearch k=%d, %d queries (stand-off cloud)" % (kk, n), lambda: tree.nearestKSearch(src, kk))
    print("   kernel ms %.2f" % tree.lastKernelMs())
q = cloud[:1_000_000]
timed("radiusSearch 1M queries r=0.002", lambda: tree.radiusSearch(q, 0.002))
ne = pcl_amd.NormalEstimation(ctx); ne.setInputCloud(cloud); ne.setSearchMethod(tree); ne.setRadiusSearch(0.002); ne.setViewPoint(0, 0, 10)
timed("normals radius 0.002 (10M)", lambda: ne.compute(want_output=False))
print("   kernel ms %.2f nan %d" % (tree.lastKernelMs(), ne.nan_count))
timed("gicp covariances k=20 (10M, host out)", lambda: tree.gicpCovariances(20), reps=1)
