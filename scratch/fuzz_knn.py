"""Randomised self-check (open-ended; tests/test_gpu_fuzz.py runs a bounded slice of the same rounds under -m gpu): k-NN,
unseeded and seeded ICP correspondences of random clouds of many shapes against the oracle, bit for bit.
python scratch/fuzz_knn.py [seed] [rounds]"""
import sys, numpy as np
sys.path.insert(0, '.')
import pcl_amd
from oracle import pcl_oracle as orc
from tests.fuzz_lib import knn_icp_round

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
ctx = pcl_amd.Context(0)
bad = 0
for it in range(rounds):
    ok, msg = knn_icp_round(ctx, orc, rng)
    bad += 0 if ok else 1
    print("it %2d %s" % (it, msg), flush=True)
print("FUZZ seed %d: %d / %d rounds with a mismatch" % (seed, bad, rounds))
sys.exit(1 if bad else 0)
