"""What would compaction buy?  (round 5, after the per-lane search was measured slower.)

The per-lane resolve pass marks the queries whose ball lies inside the kd cell of their seed's 16 / 64 / 256-point node; only
the others would have to walk the tree.  This probe measures what the EXISTING wave-cooperative kernel costs on exactly those
others, compacted in kd order into full waves: a converged alignment's moved source, the unresolved subset per level taken on
the host from pclhip_index_cells, each subset registered as a source of its own and timed in the device-driven loop
(converged launches only).  time(subset) / time(all) against the subset's share is the coherence penalty of compaction.
"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import pcl_amd
from pcl_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = pcl_amd.Context(0)
tgt = synth.gaussian_surface_device(n, synth.TARGET_SEED)
src = synth.apply_rigid_device(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface_device(n, synth.SOURCE_SEED))
tree = pcl_amd.KdTree(ctx)
tree.setInputCloud(tgt)


def converged_ms(cloud):
    icp = pcl_amd.IterativeClosestPoint(ctx)
    icp.setSearchMethodTarget(tree, True)
    icp.setInputSource(cloud)
    icp.reset()
    I = np.eye(4, dtype=np.float32)
    icp.iterate(I, max_dist=0.1)                 # the launch without seeds
    warm = []
    for _ in range(6):                           # seeded launches of a cloud that does not move: the converged state
        icp.iterate(I, max_dist=0.1)
        warm.append(icp.lastSearchMs())
    return float(np.median(warm)), icp


# the aligned state by construction: the ground-truth transform (point-to-point ICP would take tens of iterations to stop sliding)
moved = synth.apply_rigid_device(synth.ground_truth_transform(), src)
t_all, icp2 = converged_ms(moved)
print("all %d queries: converged seeded launch %.3f ms" % (n, t_all), flush=True)
# the matches of the converged state, as positions of the kd order
icp2.reset()
icp2.iterate(np.eye(4, dtype=np.float32), max_dist=0.1)
icp2.iterate(np.eye(4, dtype=np.float32), max_dist=0.1)
q, m, d2 = icp2.fetchCorrespondences()
order = tree.order()
pos_of = np.empty(len(order), np.int64)
pos_of[order] = np.arange(len(order))
pos = pos_of[m]
mv = moved.cpu().numpy()
P = mv[q, :3]
for level in (0, 1, 2, 3):
    boxes, cells, top = tree.cells(level)
    node = (pos >> 4) >> (2 * level)
    lo, hi = cells[node, :3], cells[node, 3:]
    g = np.minimum(P - lo, hi - P).min(axis=1)
    inside = (g > 0) & (g.astype(np.float32) * g.astype(np.float32) > d2)
    un = q[~inside]
    sub = torch.from_numpy(np.ascontiguousarray(mv[np.sort(un)])).cuda()
    t_sub, _ = converged_ms(sub)
    share = len(un) / len(q)
    print("level %d (%4d-point nodes): unresolved %.3f of the queries; the wave-cooperative kernel on them alone %.3f ms = %.2f of "
          "the full launch -> coherence penalty %.2fx" % (level, 16 * 4 ** level, share, t_sub, t_sub / t_all, t_sub / t_all / share), flush=True)
