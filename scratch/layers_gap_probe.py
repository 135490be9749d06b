"""Does the 'layers' family cost what it costs because its 16-point leaves straddle the two sheets?  The same cloud with the
sheets 3 / 10 / 30 / 100 point spacings apart (the kd build separates them once the gap is the widest extent of a cell)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import pcl_amd
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = pcl_amd.Context(0)
Tinv = np.linalg.inv(synth.ground_truth_transform())
for gap_sp in (0.0, 3.0, 10.0, 30.0, 100.0):
    clouds = []
    for seed in (synth.TARGET_SEED, synth.SOURCE_SEED):
        c = synth.gaussian_surface(n, seed)
        if gap_sp > 0:
            i = np.arange(n)
            c[(i & 1).astype(bool), 2] += np.float32(gap_sp * 2.0 / np.sqrt(n / 2.0))
        clouds.append(c)
    tgt = torch.from_numpy(clouds[0]).cuda()
    src = torch.from_numpy(synth.apply_rigid(Tinv, clouds[1])).cuda()
    tree = pcl_amd.KdTree(ctx); tree.setInputCloud(tgt)
    ne = pcl_amd.NormalEstimation(ctx); ne.setInputCloud(tgt); ne.setSearchMethod(tree); ne.setKSearch(8); ne.setViewPoint(0, 0, 10)
    ne.compute(want_output=False)
    icp = pcl_amd.IterativeClosestPointWithNormals(ctx)
    icp.setSearchMethodTarget(tree, True); icp.setInputSource(src)
    icp.setMaximumIterations(20); icp.setMaxCorrespondenceDistance(0.1); icp.setTransformationEpsilon(1e-10)
    icp.runSteps(5)
    st = icp.runSteps(24)
    its = {}
    for s in st:
        its.setdefault(s["iteration"], []).append(s["search_ms"])
    print("gap %5.1f spacings: step %.3f ms  search per iteration %s" % (
        gap_sp, float(np.mean([s["step_ms"] for s in st])), [round(float(np.mean(v)), 3) for k, v in sorted(its.items())][:7]), flush=True)
    del icp, ne, tree, tgt, src
