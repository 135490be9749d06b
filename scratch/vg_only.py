"""VoxelGrid(0.01) of a 10M-point surface cloud a few times (for rocprofv3 --kernel-trace)."""
import sys, time, torch
sys.path.insert(0, ".")
import pcl_amd
import pcl_amd.api as A
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = A.Context()
tgt = synth.gaussian_surface_device(n, seed=1)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vg = pcl_amd.VoxelGrid(ctx)
    vg.setInputCloud(tgt)
    vg.setLeafSize(0.01, 0.01, 0.01)
    out = vg.filter()
    ctx.synchronize()
    print("voxelgrid %.3f ms -> %d points" % ((time.perf_counter() - t0) * 1e3, out.shape[0]))
