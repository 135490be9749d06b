"""Config 3's correspondence check (tests/test_gpu_fullsize.py: cold iteration + seeded ones, every correspondence --
index AND float distance -- against the oracle's) at a size the CPU emulation of the test tier finishes in minutes.

  make -C tests/wavesim -j && PCLHIP_ALLOW_WAVESIM=1 PCLHIP_LIB=tests/wavesim/libpclhip_wavesim.so \
      python scratch/wavesim_icp_check.py 1000000

On the GPU box the same script runs against libpclhip.so (no environment needed).
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcl_amd  # noqa: E402
from oracle import pcl_oracle as orc  # noqa: E402
from pcl_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tgt = synth.gaussian_surface(n, synth.TARGET_SEED)
src = synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface(n, synth.SOURCE_SEED))
ctx = pcl_amd.Context(0)
ctx.stats(True)
t0 = time.time()
tree = pcl_amd.KdTree(ctx)
tree.setInputCloud(tgt)
print("index build %.1f s" % (time.time() - t0), flush=True)
otree = orc.KdTree(tgt)
onrm, nan = otree.normals(tgt, 8, viewpoint=(0, 0, 10))
ref = orc.icp_align(otree, tgt, src, mode=1, tgt_normals=onrm, record=True, max_iterations=20,
                    max_correspondence_distance=0.1, transformation_epsilon=1e-10)
print("oracle: %d iterations" % ref["iterations"], flush=True)
icp = pcl_amd.IterativeClosestPointWithNormals(ctx)
icp.setSearchMethodTarget(tree, True)
icp.setTargetNormals(onrm)
icp.setInputSource(src)
icp.reset()
T_prev = np.eye(4, dtype=np.float32)
cur = src.copy()
bad_total = 0
for it in range(min(ref["iterations"], 4)):
    t0 = time.time()
    sums = icp.iterate(T_prev, max_dist=0.1)
    s1 = ctx.stats(True)   # read and re-armed: the counters of this launch
    cur = orc.transform_cloud(T_prev, cur, order=1)
    oq, om, od = otree.correspondences(cur, 0.1)
    q, m, d = icp.fetchCorrespondences()
    same = len(q) == len(oq) and np.array_equal(q, oq) and np.array_equal(m, om) and \
        np.array_equal(d.view(np.uint32), od.view(np.uint32))
    nbad = -1 if len(q) != len(oq) else int((m != om).sum() + (d.view(np.uint32) != od.view(np.uint32)).sum())
    print("iteration %d: %d correspondences, identical to the oracle's: %s (%d differences), %.1f s, work counters %s" %
          (it, len(q), same, nbad, time.time() - t0, s1), flush=True)
    bad_total += 0 if same else 1
    assert np.abs(icp.solve(sums) - ref["per_iter_T"][it]).max() < 1e-6, it
    T_prev = ref["per_iter_T"][it]
# the device's own normals and the free-running (device-driven) alignment
ne = pcl_amd.NormalEstimation(ctx)
ne.setInputCloud(tgt)
ne.setSearchMethod(tree)
ne.setKSearch(8)
ne.setViewPoint(0, 0, 10)
nrm = ne.compute()
dots = np.abs(np.sum(nrm[:, :3].astype(np.float64) * onrm[:, :3].astype(np.float64), axis=1))
print("normals: min |n . n_oracle| = %.9f, nan %d" % (dots.min(), ne.nan_count), flush=True)
icp2 = pcl_amd.IterativeClosestPointWithNormals(ctx)
icp2.setSearchMethodTarget(tree, True)
icp2.setInputSource(src)
icp2.setMaximumIterations(20)
icp2.setMaxCorrespondenceDistance(0.1)
icp2.setTransformationEpsilon(1e-10)
icp2.align()
err = float(np.linalg.norm(icp2.getFinalTransformation().astype(np.float64) - ref["T"].astype(np.float64)))
print("align: %d iterations (oracle %d), |T - T_oracle|_F = %.3g" % (icp2.nr_iterations_, ref["iterations"], err))
assert bad_total == 0 and dots.min() > 1 - 1e-5 and icp2.nr_iterations_ == ref["iterations"] and err < 1e-5
print("OK")
