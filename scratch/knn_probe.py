"""nearestKSearch kernel times at n points: self-queries and the bench's stand-off cloud, k = 1 and 8 (PCLHIP_LIB picks the build)."""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
import pcl_amd
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ks = [int(a) for a in sys.argv[2:]] or [1, 8]
ctx = pcl_amd.Context(0)
cloud = torch.from_numpy(synth.gaussian_surface(n, synth.TARGET_SEED)).cuda()
src = torch.from_numpy(synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface(n, synth.SOURCE_SEED))).cuda()
tree = pcl_amd.KdTree(ctx)
tree.setInputCloud(cloud)
print(os.environ.get("PCLHIP_LIB", "libpclhip.so"))
for k in ks:
    for name, q in (("self", cloud), ("stand-off", src)):
        ms = []
        for _ in range(3):
            tree.nearestKSearch(q, k)
            ms.append(tree.lastKernelMs())
        print("k=%d %-9s kernel ms %s" % (k, name, " ".join("%.3f" % m for m in ms)))
