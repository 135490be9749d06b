"""What one per-point call of the search virtuals costs at the C ABI (pclhip_knn / pclhip_radius_search with ONE host query):
the call pattern of PCL's per-point loops over a search backend (impl/correspondence_estimation.hpp:163-175)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import pcl_amd
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tgt, src, _ = synth.icp_pair(n)
ctx = pcl_amd.Context(0)
tree = pcl_amd.KdTree(ctx)
tree.setInputCloud(tgt)
h = 2.0 / np.sqrt(n)
for k in (1, 8, 32, 33):
    q = np.ascontiguousarray(src[:2000])
    tree.nearestKSearch(q[:1], k)
    t0 = time.perf_counter()
    for i in range(len(q)):
        tree.nearestKSearch(q[i:i + 1], k)
    print("nearestKSearch(point, k=%d): %.1f us per call (k <= 32: the pinned-block path)" % (k, (time.perf_counter() - t0) / len(q) * 1e6), flush=True)
for nq in (64, 65):
    q = np.ascontiguousarray(src[:nq])
    t0 = time.perf_counter()
    for i in range(300):
        tree.nearestKSearch(q, 1)
    print("nearestKSearch(%d points, k=1): %.1f us per call" % (nq, (time.perf_counter() - t0) / 300 * 1e6), flush=True)
q = np.ascontiguousarray(src[:500])
t0 = time.perf_counter()
for i in range(len(q)):
    tree.radiusSearch(q[i:i + 1], 3 * h)
print("radiusSearch(point, 3 spacings): %.1f us per call" % ((time.perf_counter() - t0) / len(q) * 1e6), flush=True)
