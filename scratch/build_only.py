"""Build the index of a 10M-point surface cloud a few times (for rocprofv3 --kernel-trace --stats)."""
import sys, time, torch
sys.path.insert(0, ".")
import pcl_amd.api as A
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = A.Context()
tgt = synth.gaussian_surface_device(n, seed=1)
t = A.KdTree(ctx)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t.setInputCloud(tgt)
    torch.cuda.synchronize(); print("build %.3f ms" % ((time.perf_counter() - t0) * 1e3))
