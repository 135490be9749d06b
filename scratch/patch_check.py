"""A 2^20-point PATCH of the source against the 10M-point target: correspondences of host-driven launches against the oracle,
for the patch in ring order and in shuffled order (round 6: the device loop reported different pair counts and 600x
different search times for the two)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import pcl_amd
from pcl_amd import synth
from oracle import pcl_oracle as orc
n = 10_000_000
ctx = pcl_amd.Context(0)
tgt = synth.gaussian_surface_device(n, synth.TARGET_SEED)
src_all = synth.apply_rigid_device(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface_device(1 << 22, synth.SOURCE_SEED))
tree = pcl_amd.KdTree(ctx); tree.setInputCloud(tgt)
otree = orc.KdTree(tgt.cpu().numpy())
c = src_all[:, 0].abs() + src_all[:, 1].abs()
order = torch.argsort(c)
ns = 1 << 20
for shuffle in (False, True):
    idx = order[:ns]
    if shuffle:
        g = torch.Generator(device="cpu"); g.manual_seed(1)
        idx = idx[torch.randperm(ns, generator=g).to(idx.device)]
    src = src_all[idx].contiguous()
    src_h = src.cpu().numpy()
    print("finite:", bool(np.isfinite(src_h).all()), "unique rows:", len(np.unique(src_h[:, :3], axis=0)))
    icp = pcl_amd.IterativeClosestPoint(ctx)
    icp.setSearchMethodTarget(tree, True); icp.setInputSource(src); icp.reset()
    T = np.eye(4, dtype=np.float32)
    cur = src_h.copy()
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sums = icp.iterate(T, max_dist=0.1)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        cur = orc.transform_cloud(T, cur, order=0)
        q, m, d = icp.fetchCorrespondences()
        oq, om, od = otree.correspondences(cur, 0.1)
        same = len(q) == len(oq) and np.array_equal(q, oq) and np.array_equal(m, om)
        print("shuffled" if shuffle else "ring order", "launch", it, "%.2f ms" % (dt * 1e3), "pairs gpu", len(q), "oracle", len(oq), "equal", same, "count in sums", int(sums[28]), flush=True)
        T = icp.solve(sums)
    del icp
