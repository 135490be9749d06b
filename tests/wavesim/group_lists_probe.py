"""Eight host-driven ICP iterations against the oracle with the work counters of every search launch (tests/test_wavesim.py
runs it on a -DPCLHIP_GROUP_LISTS=1 build of the emulation; on the GPU box it runs against any build named by PCLHIP_LIB).
argv: points.  Prints one line per iteration and "GROUP_LISTS from_record=<share of groups of the last iteration searched from
their record> nodes=<node scans per group> rounds=<evaluation rounds per group> mismatches=<iterations that differ>"."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pcl_amd  # noqa: E402
from oracle import pcl_oracle as orc  # noqa: E402
from pcl_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
tgt, src, _ = synth.icp_pair(n)
ctx = pcl_amd.Context(0)
ctx.stats(True)
tree = pcl_amd.KdTree(ctx)
tree.setInputCloud(tgt)
otree = orc.KdTree(tgt)
onrm, _ = otree.normals(tgt, 8, viewpoint=(0, 0, 10))
icp = pcl_amd.IterativeClosestPointWithNormals(ctx)
icp.setSearchMethodTarget(tree, True)
icp.setTargetNormals(onrm)
icp.setInputSource(src)
icp.reset()
T = np.eye(4, dtype=np.float32)
cur = src.copy()
bad = 0
groups = (n + 63) // 64
for it in range(8):
    sums = icp.iterate(T, max_dist=0.1)
    st = ctx.stats(True)
    cur = orc.transform_cloud(T, cur, order=1)
    oq, om, od = otree.correspondences(cur, 0.1)
    q, m, d = icp.fetchCorrespondences()
    same = np.array_equal(q, oq) and np.array_equal(m, om) and np.array_equal(d.view(np.uint32), od.view(np.uint32))
    bad += 0 if same else 1
    share = st["so_done"] / groups if it else 0.0   # (the launch without seeds counts its stand-off groups there)
    print("iteration %d: identical to the oracle's %s; per group: node scans %.2f, per-lane leaf tests %.2f, evaluation rounds "
          "%.2f; searched from the record %.1f %%" % (it, same, st["nodes"] / groups, st["leaves_group"] / groups,
                                                      st["leaves_allpairs"] / groups, 100 * share), flush=True)
    T = icp.solve(sums)
print("GROUP_LISTS from_record=%.4f nodes=%.3f rounds=%.3f mismatches=%d" %
      (share, st["nodes"] / groups, st["leaves_allpairs"] / groups, bad))
