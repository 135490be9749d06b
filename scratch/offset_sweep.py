"""The metric's alignment under other initial misalignments and distance thresholds (10M points): first-launch and per-step
cost, iterations -- looking for cliffs off the bench's 2 degrees / 0.02 offset and max_dist 0.1."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import pcl_amd
from pcl_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = pcl_amd.Context(0)
tgt = synth.gaussian_surface_device(n, synth.TARGET_SEED)
base = synth.gaussian_surface_device(n, synth.SOURCE_SEED)
tree = pcl_amd.KdTree(ctx); tree.setInputCloud(tgt)
ne = pcl_amd.NormalEstimation(ctx); ne.setInputCloud(tgt); ne.setSearchMethod(tree); ne.setKSearch(8); ne.setViewPoint(0, 0, 10)
ne.compute(want_output=False)

def rigid(deg, t):
    ax = np.array([0.3, -0.5, 0.81]); ax /= np.linalg.norm(ax)
    a = np.deg2rad(deg); K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    return T

def run(label, T, max_dist=0.1, cls=pcl_amd.IterativeClosestPointWithNormals, iters=20):
    src = synth.apply_rigid_device(np.linalg.inv(T), base)
    torch.cuda.synchronize()
    icp = cls(ctx)
    icp.setSearchMethodTarget(tree, True); icp.setInputSource(src)
    icp.setMaximumIterations(iters); icp.setMaxCorrespondenceDistance(max_dist); icp.setTransformationEpsilon(1e-10)
    icp.runSteps(3)
    st = icp.runSteps(24)
    its = {}
    for s in st:
        its.setdefault(s["iteration"], []).append(s["search_ms"])
    ends = [s["iteration"] for s in st if s["alignment_ended"]]
    print("%-44s step %.3f ms, iterations %s, pairs %d, search per iteration %s" % (
        label, float(np.mean([s["step_ms"] for s in st])), ends[:2], int(st[-1]["num_correspondences"]),
        [round(float(np.mean(v)), 3) for k, v in sorted(its.items())][:6]), flush=True)
    del icp

for scale in (0.0, 0.25, 1.0, 2.0, 4.0, 8.0):
    run("offset x%.2f (%.1f deg, |t| %.3f)" % (scale, 2 * scale, 0.0213 * scale), rigid(2.0 * scale, np.array([0.012, -0.009, 0.015]) * scale))
for md in (0.001, 0.01, 1.0, 1e9):
    run("bench offset, max_dist %g" % md, rigid(2.0, np.array([0.012, -0.009, 0.015])), max_dist=md)
run("bench offset, point-to-point (SVD)", rigid(2.0, np.array([0.012, -0.009, 0.015])), cls=pcl_amd.IterativeClosestPoint)
