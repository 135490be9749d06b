"""Randomised self-check (open-ended; tests/test_gpu_fuzz.py runs a bounded slice of the same rounds under -m gpu): VoxelGrid
(leaf sizes, pass-through filter on any field, negative limits, minimum points per voxel, PointXYZ / PointNormal records)
and NormalEstimation with a search surface / index subset against the oracle.  python scratch/fuzz_filters.py [seed] [rounds]"""
import sys, numpy as np
sys.path.insert(0, '.')
import pcl_amd
from oracle import pcl_oracle as orc
from tests.fuzz_lib import filters_round

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(seed)
ctx = pcl_amd.Context(0)
bad = 0
for it in range(rounds):
    ok, msg = filters_round(ctx, orc, rng)
    bad += 0 if ok else 1
    print("it %2d %s" % (it, msg), flush=True)
print("FUZZ filters seed %d: %d / %d rounds with a mismatch" % (seed, bad, rounds))
sys.exit(1 if bad else 0)
