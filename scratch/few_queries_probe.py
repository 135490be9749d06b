"""Batch k-NN with FEW queries against a large index: a wavefront owns 64 consecutive (sorted) queries, which are far apart
when nq << n.  Time per call by nq (random queries on the surface)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import pcl_amd
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
tgt = synth.gaussian_surface_device(n, seed=1)
qs = synth.gaussian_surface_device(1 << 20, seed=2)
ctx = pcl_amd.Context(0)
tree = pcl_amd.KdTree(ctx)
tree.setInputCloud(tgt)
for nq in (1, 16, 64, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20):
    q = qs[:nq].contiguous()
    for k in (1, 8):
        tree.nearestKSearch(q, k)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            tree.nearestKSearch(q, k)
        torch.cuda.synchronize()
        print("n %d nq %7d k %d: %9.1f us per call, %8.3f us per query" % (n, nq, k, (time.perf_counter() - t0) / reps * 1e6, (time.perf_counter() - t0) / reps * 1e6 / nq), flush=True)
