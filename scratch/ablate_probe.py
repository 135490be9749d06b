"""timing ablations of the seeded search body (variant built with the ablation switches; option lane_max_up carries the code):
an alignment with code 0, then for every code a few launches at the converged pose whose outputs are not stored"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import pcl_amd
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
upto = int(sys.argv[2]) if len(sys.argv) > 2 else 5     # iterations run normally before the ablated launches
ctx = pcl_amd.Context(0)
ctx.setOption("lane_max_up", 0)
tgt = torch.from_numpy(synth.gaussian_surface(n, synth.TARGET_SEED)).cuda()
src = torch.from_numpy(synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface(n, synth.SOURCE_SEED))).cuda()
tree = pcl_amd.KdTree(ctx); tree.setInputCloud(tgt)
ne = pcl_amd.NormalEstimation(ctx); ne.setInputCloud(tgt); ne.setSearchMethod(tree); ne.setKSearch(8); ne.setViewPoint(0, 0, 10); ne.compute(want_output=False)
icp = pcl_amd.IterativeClosestPointWithNormals(ctx)
icp.setSearchMethodTarget(tree); icp.setInputSource(src)
icp.reset(); T = np.eye(4, dtype=np.float32); ms = []
for it in range(upto):
    sums = icp.iterate(T, max_dist=0.1); T = icp.solve(sums); ms.append(icp.lastKernelMs())
print("normal launches:", " ".join("%.3f" % m for m in ms))
I = np.eye(4, dtype=np.float32)
names = {0: "full", 1: "full, outputs not stored", 2: "no resolve / index gather / tie pass", 3: "no traversal at all (loads, transform, seed distance)",
         4: "traversal without evaluation rounds (lists, box tests, staging)", 5: "traversal stops after the start-level test",
         6: "node visits + lists only (no box tests, staging, rounds)", 7: "node visits + lists + box tests (no staging, no rounds)"}
for code in [1, 2, 3, 4, 5, 6, 7, 1]:
    ctx.setOption("lane_max_up", code)
    t = []
    for r in range(4):
        icp.iterate(I if upto >= 4 else T, max_dist=0.1); t.append(icp.lastKernelMs())
    print("code %d  %-68s ms %s" % (code, names[code], " ".join("%.3f" % x for x in t)))
