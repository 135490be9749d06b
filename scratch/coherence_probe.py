"""Probe (round 6): how far do the queries move between consecutive ICP iterations, and for what share of them is the
previous match still provably the nearest point by the second-nearest-distance gap?  Oracle only (CPU); config 3 geometry.
usage: python scratch/coherence_probe.py [n]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import pcl_amd.synth as synth
from oracle import pcl_oracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
tgt, src, T_gt = synth.icp_pair(n)
t0 = time.time()
tree = orc.KdTree(tgt)
nrm, _ = tree.normals(tgt, 8, viewpoint=(0, 0, 10))
print("tree+normals %.1fs" % (time.time() - t0), flush=True)
res = orc.icp_align(tree, tgt, src, mode=1, tgt_normals=nrm, record=True, max_iterations=20,
                    max_correspondence_distance=0.1, transformation_epsilon=1e-10)
Ts = res["per_iter_T"]  # T applied before iteration k's search? (cumulative)
print("iterations", res["iterations"])
h = 2.0 / np.sqrt(n)
print("spacing h = %.3g" % h)
cum = np.eye(4, dtype=np.float64)
pos_prev = None
sub = slice(0, min(n, 400_000))
for k in range(res["iterations"]):
    cum = Ts[k].astype(np.float64) @ cum  # per_iter_T holds the INCREMENTAL transformation_ of iteration k
    T = cum
    # positions the search of iteration k+1 saw = cumulative transform after k incremental ones: try to detect convention
    pos = src[sub, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    if pos_prev is not None:
        d = np.linalg.norm(pos - pos_prev, axis=1)
        idx, d2 = tree.knn(np.concatenate([pos_prev.astype(np.float32), np.ones((len(pos), 1), np.float32)], 1), 2)
        gap = np.sqrt(d2[:, 1]) - np.sqrt(d2[:, 0])
        # certificate: sqrt(s) - delta > dist(q', m)  with dist(q',m) <= sqrt(d0) + delta  ->  gap > 2 delta suffices
        ok = gap > 2 * d * 1.0001 + 1e-7
        # exact version: new distance to the old match
        pm = tgt[idx[:, 0], :3].astype(np.float64)
        dnew = np.linalg.norm(pos - pm, axis=1)
        ok2 = np.sqrt(d2[:, 1]) - d > dnew * 1.00001
        print("iter %d -> %d: move mean %.3g (%.2f h) max %.3g | gap median %.3g (%.2f h) | certified (2-NN gap) %.2f%%  exact form %.2f%%"
              % (k, k + 1, d.mean(), d.mean() / h, d.max(), np.median(gap), np.median(gap) / h, 100 * ok.mean(), 100 * ok2.mean()), flush=True)
    pos_prev = pos
q = np.concatenate([pos_prev.astype(np.float32), np.ones((len(pos_prev), 1), np.float32)], 1)
idx, d2 = tree.knn(q, 3)
print("d1 d2 d3 medians (in h):", np.median(np.sqrt(d2), axis=0) / h)
