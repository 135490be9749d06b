"""The stand-off body beyond the old 640 MB gate: the launch that starts an alignment (and two seeded ones) at 20M points
against the oracle, every correspondence (index and float distance)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import pcl_amd
from pcl_amd import synth
from oracle import pcl_oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
ctx = pcl_amd.Context(0)
tgt = synth.gaussian_surface_device(n, synth.TARGET_SEED)
src = synth.apply_rigid_device(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface_device(n, synth.SOURCE_SEED))
torch.cuda.synchronize()
tree = pcl_amd.KdTree(ctx); tree.setInputCloud(tgt)
tgt_h, src_h = tgt.cpu().numpy(), src.cpu().numpy()
otree = orc.KdTree(tgt_h)
icp = pcl_amd.IterativeClosestPoint(ctx)
icp.setSearchMethodTarget(tree, True); icp.setInputSource(src); icp.reset()
T = np.eye(4, dtype=np.float32); cur = src_h.copy()
for it in range(3):
    sums = icp.iterate(T, max_dist=0.1)
    cur = orc.transform_cloud(T, cur, order=0)
    oq, om, od = otree.correspondences(cur, 0.1)
    q, m, d = icp.fetchCorrespondences()
    ok = np.array_equal(q, oq) and np.array_equal(m, om) and np.array_equal(d.view(np.uint32), od.view(np.uint32))
    print("%d points, launch %d: %d correspondences, equal to the oracle's: %s" % (n, it, len(q), ok), flush=True)
    assert ok
    T = icp.solve(sums)
