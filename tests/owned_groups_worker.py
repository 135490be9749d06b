"""One run of the served-group scenario of tests/test_gpu_dist.py::test_served_group_lists_equal_the_full_pass (a process
of its own so that both runs start from a fresh context).  argv: out.npz points served_groups(0|1)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcl_amd  # noqa: E402
from pcl_amd import synth  # noqa: E402

out, n = sys.argv[1], int(sys.argv[2])
tgt, src, _ = synth.icp_pair(n)
# a start 0.06 off along x on top of the pair's own 2 degrees: the cloud sweeps across the strips below while it converges
guess = np.eye(4, dtype=np.float32)
guess[0, 3] = 0.06
ctx = pcl_amd.Context(0)
ctx.setOption("served_groups", int(sys.argv[3]))   # 0: every launch walks the whole source (the reference of the test)
tree = pcl_amd.KdTree(ctx)
tree.setInputCloud(tgt)
ne = pcl_amd.NormalEstimation(ctx)
ne.setInputCloud(tgt)
ne.setSearchMethod(tree)
ne.setKSearch(8)
ne.setViewPoint(0, 0, 10)
ne.compute(want_output=False)
res = {}
inf = np.inf
regions = {"strip": [0.10, -inf, -inf, 0.22, inf, inf],            # a strip the cloud moves through
           "corner": [-inf, 0.35, -inf, -0.2, inf, inf],           # a corner piece
           "all": [-inf] * 3 + [inf] * 3}
for name, region in regions.items():
    for mode, cls in (("plane", pcl_amd.IterativeClosestPointWithNormals), ("point", pcl_amd.IterativeClosestPoint)):
        icp = cls(ctx)
        icp.setSearchMethodTarget(tree, True)
        icp.setInputSource(src)
        icp.setMaximumIterations(12)
        icp.setMaxCorrespondenceDistance(0.1)
        icp.setTransformationEpsilon(1e-10)
        icp.setRegion(region)
        icp.align(guess)
        q, m, d = icp.fetchCorrespondences()
        key = name + "_" + mode
        res[key + "_T"] = icp.getFinalTransformation().copy()
        res[key + "_iterations"] = icp.nr_iterations_
        res[key + "_q"], res[key + "_m"], res[key + "_d"] = q, m, d
        # a host-driven iteration on top of the finished alignment: it moves the WHOLE working cloud, also the groups the
        # loop's last launches did not serve
        nudge = np.eye(4, dtype=np.float32)
        nudge[0, 3] = -0.03
        icp.iterate(nudge, max_dist=0.1)
        q, m, d = icp.fetchCorrespondences()
        res[key + "_q3"], res[key + "_m3"], res[key + "_d3"] = q, m, d
        steps = icp.runSteps(15)   # whole alignments back to back (from the identity): restarts inside the queue
        res[key + "_counts"] = np.asarray([s["num_correspondences"] for s in steps], np.float64)
        res[key + "_mse"] = np.asarray([s["mse"] for s in steps], np.float64)
        res[key + "_its"] = np.asarray([s["iteration"] for s in steps])
        q, m, d = icp.fetchCorrespondences()
        res[key + "_q2"], res[key + "_m2"], res[key + "_d2"] = q, m, d
np.savez(out, **res)
