"""A/B of the per-lane seeded search in ONE process: the bench's config-3 loop (10M-point clouds by default) under several
option sets of pclhip_ctx_set_option, one context each; prints per-iteration search times, ms/step and the lane counters.
  python scratch/lane_ab.py [n] [cloud] "lane_search=0" "lane_max_up=1" "lane_max_up=3,lane_far=1" ...
"""
import collections
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import pcl_amd
from pcl_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
cloud = sys.argv[2] if len(sys.argv) > 2 else "sheet"
sets = sys.argv[3:] or [""]
t0 = time.time()
if cloud == "sheet":
    tgt = synth.gaussian_surface_device(n, synth.TARGET_SEED)
    src = synth.apply_rigid_device(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface_device(n, synth.SOURCE_SEED))
else:
    tgt = torch.from_numpy(synth.family_cloud(cloud, n, synth.TARGET_SEED)).cuda()
    src = torch.from_numpy(synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()),
                                             synth.family_cloud(cloud, n, synth.SOURCE_SEED))).cuda()
torch.cuda.synchronize()
print("clouds %.1f s (%s, %d points)" % (time.time() - t0, cloud, n), flush=True)
for opts in sets:
    ctx = pcl_amd.Context(0)
    for kv in [o for o in opts.split(",") if o]:
        name, _, value = kv.partition("=")
        ctx.setOption(name, float(value))
    tree = pcl_amd.KdTree(ctx)
    tree.setInputCloud(tgt)
    tree.setInputCloud(tgt)
    build_ms = tree.build_ms()
    ne = pcl_amd.NormalEstimation(ctx)
    ne.setInputCloud(tgt)
    ne.setSearchMethod(tree)
    ne.setKSearch(8)
    ne.setViewPoint(0, 0, 10)
    ne.compute(want_output=False)
    icp = pcl_amd.IterativeClosestPointWithNormals(ctx)
    icp.setSearchMethodTarget(tree, True)
    icp.setInputSource(src)
    icp.setMaximumIterations(20)
    icp.setMaxCorrespondenceDistance(0.1)
    icp.setTransformationEpsilon(1e-10)
    icp.runSteps(5)
    ctx.synchronize()
    t1 = time.perf_counter()
    steps = icp.runSteps(20)
    ctx.synchronize()
    el = time.perf_counter() - t1
    by = collections.defaultdict(list)
    for s in steps:
        by[s["iteration"]].append(s["search_ms"])
    ov = [s["step_ms"] - s["search_ms"] for s in steps]
    line = "[%s] ms_per_step %.4f search/iter %s rest %.3f build %.2f" % (
        opts or "default", el / 20 * 1e3, " ".join("%d:%.3f" % (k, sum(v) / len(v)) for k, v in sorted(by.items())),
        sum(ov) / len(ov), build_ms)
    # the lane counters launch by launch: one alignment replayed host-driven with the device loop's own transforms
    one = icp.runSteps(len(by))
    icp.reset()
    prev = np.eye(4)
    T = np.eye(4, dtype=np.float32)
    for k, s in enumerate(one):
        if k > 0:
            ctx.counters(True)
        icp.iterate(T, max_dist=0.1)
        if k > 0:
            c = ctx.counters(False)
            if c[0]:
                line += " | it%d own-leaf %.3f pass1 %.3f greedy %.3f" % (k + 1, c[1] / c[0], c[2] / c[0], c[3] / c[0])
        F = s["final_transformation"].astype(np.float64)
        T = (F @ np.linalg.inv(prev)).astype(np.float32)
        prev = F
    print(line, flush=True)
    del icp, ne, tree
    ctx.close()
