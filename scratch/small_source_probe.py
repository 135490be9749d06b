"""ICP with a SMALL source against a large target (a scan against a map): ms per iteration by source size, for a source
spread over the whole map and for one that covers a patch of it.  (Device tensors are synchronised before they are handed
over: the context works on its own stream.)"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import pcl_amd
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = pcl_amd.Context(0)
tgt = synth.gaussian_surface_device(n, synth.TARGET_SEED)
src_all = synth.apply_rigid_device(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface_device(1 << 22, synth.SOURCE_SEED))
tree = pcl_amd.KdTree(ctx); tree.setInputCloud(tgt)
ne = pcl_amd.NormalEstimation(ctx); ne.setInputCloud(tgt); ne.setSearchMethod(tree); ne.setKSearch(8); ne.setViewPoint(0, 0, 10)
ne.compute(want_output=False)
order = torch.argsort(src_all[:, 0].abs() + src_all[:, 1].abs())
for ns in (256, 4096, 65536, 1 << 20, 1 << 22):
    for local in (False, True):
        src = (src_all[order[:ns]] if local else src_all[:ns]).contiguous()
        torch.cuda.synchronize()
        icp = pcl_amd.IterativeClosestPointWithNormals(ctx)
        icp.setSearchMethodTarget(tree, True); icp.setInputSource(src)
        icp.setMaximumIterations(20); icp.setMaxCorrespondenceDistance(0.1); icp.setTransformationEpsilon(1e-10)
        icp.runSteps(5)
        st = icp.runSteps(20)
        its = {}
        for s in st:
            its.setdefault(s["iteration"], []).append(s["search_ms"])
        print("source %8d points (%s): step %.3f ms, %d pairs, search per iteration %s" % (
            ns, "one patch" if local else "spread over the map", float(np.mean([s["step_ms"] for s in st])), int(st[-1]["num_correspondences"]),
            [round(float(np.mean(v)), 3) for k, v in sorted(its.items())][:6]), flush=True)
        del icp
