"""What would perfect seeds buy?  Every launch of an alignment is run twice from the same pose: as it comes (seeds = the
previous launch's matches) and again with the identity transform (seeds = its OWN matches: the lower bound of any seeding
improvement).  Host-driven, 10M points."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import pcl_amd
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = pcl_amd.Context(0)
tgt = synth.gaussian_surface_device(n, synth.TARGET_SEED)
src = synth.apply_rigid_device(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface_device(n, synth.SOURCE_SEED))
torch.cuda.synchronize()
tree = pcl_amd.KdTree(ctx); tree.setInputCloud(tgt)
ne = pcl_amd.NormalEstimation(ctx); ne.setInputCloud(tgt); ne.setSearchMethod(tree); ne.setKSearch(8); ne.setViewPoint(0, 0, 10)
ne.compute(want_output=False)
I = np.eye(4, dtype=np.float32)
for cls, iters in ((pcl_amd.IterativeClosestPointWithNormals, 5), (pcl_amd.IterativeClosestPoint, 10)):
    icp = cls(ctx)
    icp.setSearchMethodTarget(tree, True); icp.setInputSource(src); icp.reset()
    T = I
    for it in range(iters):
        sums = icp.iterate(T, max_dist=0.1)
        t_real = icp.lastSearchMs()
        icp.iterate(I, max_dist=0.1)
        t_perfect = icp.lastSearchMs()
        Tn = icp.solve(sums)
        step = float(np.linalg.norm(Tn[:3, 3])) + float(np.linalg.norm(Tn[:3, :3] - np.eye(3)))
        print("%s launch %d: %.3f ms as it comes, %.3f ms with its own matches as seeds (next transform moves by ~%.2g = %.1f spacings)" % (
            cls.__name__, it + 1, t_real, t_perfect, step, step / (2 / np.sqrt(n))), flush=True)
        T = Tn
    del icp
