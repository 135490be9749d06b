"""30 steps of the device-driven loop (whole alignments back to back, both estimators) as one JSON line: iterations, states,
counts, MSEs to the last bit and checksums of the final correspondences.  tests/test_wavesim.py compares the line of a variant
build with the default build's (the group leaf lists keep records across the restarts inside the queue).  argv: points"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pcl_amd
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
tgt, src, _ = synth.icp_pair(n)
ctx = pcl_amd.Context(0)
tree = pcl_amd.KdTree(ctx); tree.setInputCloud(tgt)
ne = pcl_amd.NormalEstimation(ctx); ne.setInputCloud(tgt); ne.setSearchMethod(tree); ne.setKSearch(8); ne.setViewPoint(0,0,10); ne.compute(want_output=False)
out={}
for name, cls in (("plane", pcl_amd.IterativeClosestPointWithNormals), ("point", pcl_amd.IterativeClosestPoint)):
    icp = cls(ctx); icp.setSearchMethodTarget(tree, True); icp.setInputSource(src); icp.setMaximumIterations(20); icp.setMaxCorrespondenceDistance(0.1); icp.setTransformationEpsilon(1e-10)
    steps = icp.runSteps(30)
    out[name] = [(s["iteration"], s["state"], int(s["num_correspondences"]), repr(float(s["mse"])), bool(s["alignment_ended"])) for s in steps]
    q,m,d = icp.fetchCorrespondences()
    out[name+"_corr"] = [int(q.sum()), int(m.astype(np.int64).sum()), repr(float(d.astype(np.float64).sum()))]
print(json.dumps(out))
