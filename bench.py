#!/usr/bin/env python3
"""bench.py -- the metric of BASELINE.json ("ICP correspondences/sec/GPU + ms/iteration, 10M-pt cloud; HBM
GB/s vs roofline") on N MI355X GPUs of one node.

  --config 3 (default)  10M-point synthetic Gaussian-surface clouds, k = 8 NormalEstimation + point-to-plane ICP
  --config 2            2^20-point clouds, k = 1, point-to-point ICP (TransformationEstimationSVD)
  --config 4            config 3's clouds through VoxelGrid(0.01) first; a step is the whole pipeline
                        (filter both clouds + target index + normals + the alignment), as SURVEY.md 8(d) says
  --config 5            100M-point target cut into kd slabs + halo over the ranks (needs --gpus > 1; the
                        replicated-target variant of the same problem is --config 5 --replicated)

Configs 2/3/5: a "step" is ONE ICP iteration over the whole (rank-local) source cloud: transform -> exact 1-NN
in the target -> normal-system accumulation -> (all-reduce) -> solve + convergence test.  The K timed steps are
queued back to back by pclhip_icp_run_steps -- the iteration is closed on the device, nothing returns to the
host in between -- and come from repeated IterativeClosestPoint::align() runs on the config's clouds (when an
alignment converges the next one starts from the input cloud), so they contain the realistic mix of
far-from-aligned and nearly-aligned iterations; the per-step list in the output says which was which.
Target index + normals are built before the timed region (reported under "setup") and stay resident in HBM.

Multi-GPU (weak scaling, configs 2/3): every rank holds the full target index and its own slab of the source;
the only exchange per iteration is the all-reduce of the 32-double record (ncclAllReduce from C, RCCL/xGMI).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic bytes per unit; DESIGN.md section 4 repeats the derivations
B_ALG_SEARCH = 40.0                     # src 16 + matched tgt 16 + (idx 4 + d2 4)
B_ALG_ITER = {1: 56.0, 0: 40.0}         # + target normal 16 for point-to-plane
B_ALG_VOXEL = 32.0                      # per input point
B_ALG_NORMALS = 160.0                   # per target point, k = 8
B_ALG_BUILD = 64.0                      # per indexed point
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4, 5])
    ap.add_argument("--points", type=int, default=0, help="override the config's points per cloud (per GPU)")
    ap.add_argument("--knn", type=int, default=8, help="k of NormalEstimation")
    ap.add_argument("--replicated", action="store_true", help="config 5: replicate the target instead of sharding it")
    ap.add_argument("--virtual-world", type=int, default=0,
                    help="config 5 on ONE GPU: play rank --virtual-rank of this many ranks (its kd slab + halo, its region "
                         "mask, no collective): what one rank of a G-GPU job does per iteration, measurable without the node")
    ap.add_argument("--virtual-rank", type=int, default=0)
    ap.add_argument("--full-pass", action="store_true",
                    help="config 5: every launch walks the whole source (option served_groups = 0) instead of the served groups")
    ap.add_argument("--rejectors", default="", help="comma list of median,trimmed,one_to_one,distance: the rejector chain "
                                                     "inside the device-driven loop (configs 2/3)")
    ap.add_argument("--reciprocal", action="store_true", help="reciprocal correspondences inside the device-driven loop")
    ap.add_argument("--cloud", default="sheet", choices=["sheet", "cube", "layers", "clusters"],
                    help="cloud family of configs 2/3 (pcl_amd/synth.py: family_cloud); the metric's config is the sheet")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="pclhip_ctx_set_option before anything is built (A/B runs: lane_search=0, lane_max_up=1, ...)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-families", action="store_true",
                    help="skip the `families` sub-record (the same measurement on the cube / layers / clusters clouds)")
    ap.add_argument("--no-host-align", action="store_true", help="skip the host-boundary timing (examples/bench_pcl_align.cpp)")
    return ap.parse_args()


def kernel_sources_sha():
    """sha256 over the sources of the search kernel and of the index build that shapes the tree it walks (scripts/make_pmc_traffic.py
    stamps the counter passes with it)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("search.hip", "traverse.hpp", "pclhip_wave_reduce.hpp", "standoff.hpp", "pclhip_internal.hpp", "index_build.hip"):
        h.update(open(os.path.join(ROOT, "pcl_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def git_head():
    if os.environ.get("GRAFT_COMMIT"):  # the GPU box runs a snapshot without .git: the launcher passes the commit
        return os.environ["GRAFT_COMMIT"]
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL,
                                       text=True).strip()
    except Exception:
        return None


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import pcl_amd
    from pcl_amd import synth
    from pcl_amd.dist import init_ranks, make_fence, max_over_ranks, native_communicator, timed_steps

    # rendezvous, the native communicator and the fences: pcl_amd/dist.py -- the same functions the 2- and 3-process CPU
    # tests run (tests/wavesim/two_rank_worker.py over gloo and the emulation), so this control flow is not new to N > 1
    rank, local_rank, world = init_ranks("nccl")
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    cfg = args.config
    mode = 0 if cfg == 2 else 1
    n = args.points or {2: 1 << 20, 3: 10_000_000, 4: 10_000_000, 5: 100_000_000}[cfg]
    ctx = pcl_amd.Context(local_rank)      # its own stream; collectives are issued on it from C
    for kv in args.opt:
        name, _, value = kv.partition("=")
        ctx.setOption(name, float(value))
    fence = make_fence(ctx, world)
    comm = native_communicator(ctx, rank, world)   # rank 0's id travels over the torch process group once

    if cfg == 5:
        from bench_sharded import run_config5     # target slab + halo sharding (pcl_amd/dist.py)
        out = run_config5(args, ctx, comm, rank, local_rank, world, fence)
        finish(out, rank, world)
        return

    # ---- synthetic clouds; target identical on every rank, source = this rank's slab
    t0 = time.perf_counter()
    tgt_h = synth.family_cloud(args.cloud, n, synth.TARGET_SEED)
    src_h = synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()),
                              synth.family_cloud(args.cloud, n, synth.SOURCE_SEED, start=rank * n))
    gen_s = time.perf_counter() - t0
    tgt = torch.from_numpy(tgt_h).cuda()
    src = torch.from_numpy(src_h).cuda()

    if cfg == 4:
        out = run_pipeline(args, ctx, comm, tgt, src, n, rank, world, fence, gen_s)
        finish(out, rank, world)
        return

    # ---- target index + normals (one-off, outside the timed region)
    tree = pcl_amd.KdTree(ctx)
    tree.setInputCloud(tgt)
    build_first_ms = tree.build_ms()    # includes the context's first hipMallocs (scratch + index arrays)
    tree.setInputCloud(tgt)             # a registration pipeline re-indexes every frame: the steady state
    build_ms = tree.build_ms()
    normals_ms = normals_first_ms = None
    if mode == 1:
        ne = pcl_amd.NormalEstimation(ctx)
        ne.setInputCloud(tgt)
        ne.setSearchMethod(tree)
        ne.setKSearch(args.knn)
        ne.setViewPoint(0, 0, 10)
        ne.compute(want_output=False)
        normals_first_ms = tree.lastKernelMs()   # the process's first launch of the kernel loads its code object
        ne.compute(want_output=False)
        normals_ms = tree.lastKernelMs()

    cls = pcl_amd.IterativeClosestPointWithNormals if mode == 1 else pcl_amd.IterativeClosestPoint
    icp = cls(ctx)
    icp.setSearchMethodTarget(tree, True)
    icp.setInputSource(src)
    icp.setMaximumIterations(20)
    icp.setMaxCorrespondenceDistance(0.1)
    icp.setTransformationEpsilon(1e-10)
    from pcl_amd.dist import attach_collective
    collective = attach_collective(icp, comm, local_rank, world)   # "native" (RCCL from C), or torch's all-reduce if that failed
    for name in [r for r in args.rejectors.split(",") if r]:
        if name == "median":
            rej = pcl_amd.CorrespondenceRejectorMedianDistance()
            rej.setMedianFactor(2.0)
        elif name == "trimmed":
            rej = pcl_amd.CorrespondenceRejectorTrimmed()
            rej.setOverlapRatio(0.9)
        elif name == "one_to_one":
            rej = pcl_amd.CorrespondenceRejectorOneToOne()
        elif name == "distance":
            rej = pcl_amd.CorrespondenceRejectorDistance()
            rej.setMaximumDistance(0.05)
        else:
            raise SystemExit("unknown rejector %r" % name)
        icp.addCorrespondenceRejector(rej)
    if args.reciprocal:
        icp.setUseReciprocalCorrespondences(True)
    source_order_ms = icp.sourceOrderMs()

    steps, elapsed = timed_steps(icp, args.steps, args.warmup, fence, world)

    # a fresh frame: a NEW source cloud against the resident target -- its ordering, then one whole alignment (device times)
    fresh_frame = None
    if world == 1 and not args.rejectors and not args.reciprocal:
        icp_f = cls(ctx)
        icp_f.setSearchMethodTarget(tree, True)
        icp_f.setInputSource(src)
        icp_f.setMaximumIterations(20)
        icp_f.setMaxCorrespondenceDistance(0.1)
        icp_f.setTransformationEpsilon(1e-10)
        tf0 = time.perf_counter()
        many = icp_f.runSteps(21)
        ctx.synchronize()
        one = []
        for s_ in many:                       # the first whole alignment of the run
            one.append(s_)
            if s_["alignment_ended"]:
                break
        fresh_frame = {"source_order_ms": round(icp_f.sourceOrderMs(), 3), "iterations": len(one),
                 "alignment_device_ms": round(sum(s_["step_ms"] for s_ in one), 3),
                 "fresh_frame_ms": round(icp_f.sourceOrderMs() + sum(s_["step_ms"] for s_ in one), 3),
                 "wall_ms_of_21_steps": round((time.perf_counter() - tf0) * 1e3, 3)}
        del icp_f
    ncorr = float(sum(s["num_correspondences"] for s in steps))   # already the all-reduced (global) count
    search_ms = sum(s["search_ms"] for s in steps)
    kernel_ms = sum(s["kernels_ms"] for s in steps)
    # ---- roofline of the dominant kernel (the exact 1-NN search), live HIP-event timing on the context's
    # stream: achieved = algorithmic bytes per launch / average launch duration
    avg_kernel_s = search_ms / max(args.steps, 1) / 1e3
    corr_per_launch_local = ncorr / max(args.steps, 1) / world
    achieved = B_ALG_SEARCH * corr_per_launch_local / avg_kernel_s / 1e9
    roofline = {"bound": "hbm",
                "kernel": "icp_search_dual_kernel (one launch per iteration: the stand-off body for the first iteration of an "
                          "alignment, the seeded body after it -- with a greedy reseed where every seed of a group is far; both "
                          "are averaged here as the steps mix them)",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": None, "alg_bytes_per_corr": B_ALG_SEARCH, "avg_kernel_ms": round(avg_kernel_s * 1e3, 4),
                "iteration_kernels_ms": round(kernel_ms / max(args.steps, 1), 4),
                "iteration_alg_bytes_per_corr": B_ALG_ITER[mode]}
    traffic_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if cfg == 3 and os.path.exists(traffic_file):
        # HBM bytes per launch come from separate rocprofv3 --pmc passes of this very command (scripts/profile_gpu.sh
        # regenerates the file and stamps the commit and the hash of the search kernel's sources it measured); a file
        # that was measured on other kernel sources than the ones in this tree is NOT reported
        try:
            tj = json.load(open(traffic_file))
            fresh = tj.get("kernel_sources_sha") == kernel_sources_sha()
            roofline["traffic"] = tj.get("icp_search_bytes_per_launch") if fresh else None
            if fresh and roofline["traffic"]:
                # SURVEY.md 8(d)'s "measured": the counters' bytes over the launch's duration, beside the algorithmic figure
                roofline["measured"] = round(roofline["traffic"] / avg_kernel_s / 1e9, 2)
                roofline["measured_frac"] = round(roofline["traffic"] / avg_kernel_s / 1e9 / HBM_PEAK_GBS, 5)
                roofline["traffic_over_algorithmic"] = round(roofline["traffic"] / (B_ALG_SEARCH * corr_per_launch_local), 3)
            roofline["traffic_source"] = {"file": "profiles/pmc_traffic.json", "commit": tj.get("commit"),
                                          "date": tj.get("date"), "command": tj.get("command"),
                                          "kernel_sources_match": fresh,
                                          "note": ("NOT measured in this run: FETCH_SIZE / WRITE_SIZE need their own "
                                                   "rocprofv3 --pmc passes; this is the per-launch average of those passes "
                                                   "of the same command at the commit named here, whose search kernel "
                                                   "sources are the ones of this tree") if fresh else
                                                  "the counter passes on file are of other kernel sources: traffic not reported"}
        except Exception:
            pass

    # ---- the same measurement OFF the metric's geometry: a volume, two close layers, 100x density contrast
    # (pcl_amd/synth.py: family_cloud; PCL's own search tests use volumetric random clouds, test/search/test_search.cpp
    # :292-364).  Not part of `value`: a sub-record, so that the driver's line shows how far the headline generalises.
    families = None
    if (cfg == 3 and world == 1 and args.cloud == "sheet" and not args.no_families and not args.rejectors and not args.reciprocal
            and not args.points):
        families = {}
        sheet_ms = elapsed / max(args.steps, 1) * 1e3
        sheet_first = float(np.mean([s_["search_ms"] for s_ in steps if s_["iteration"] == 1] or [0.0]))
        for kind in ("cube", "layers", "clusters"):
            f_tgt = torch.from_numpy(synth.family_cloud(kind, n, synth.TARGET_SEED)).cuda()
            f_src = torch.from_numpy(synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()),
                                                       synth.family_cloud(kind, n, synth.SOURCE_SEED))).cuda()
            f_tree = pcl_amd.KdTree(ctx)
            f_tree.setInputCloud(f_tgt)
            f_ne = pcl_amd.NormalEstimation(ctx)
            f_ne.setInputCloud(f_tgt)
            f_ne.setSearchMethod(f_tree)
            f_ne.setKSearch(args.knn)
            f_ne.setViewPoint(0, 0, 10)
            f_ne.compute(want_output=False)
            f_icp = cls(ctx)
            f_icp.setSearchMethodTarget(f_tree, True)
            f_icp.setInputSource(f_src)
            f_icp.setMaximumIterations(20)
            f_icp.setMaxCorrespondenceDistance(0.1)
            f_icp.setTransformationEpsilon(1e-10)
            f_steps, f_elapsed = timed_steps(f_icp, args.steps, args.warmup, fence, world)
            f_ms = f_elapsed / max(args.steps, 1) * 1e3
            f_first = float(np.mean([s_["search_ms"] for s_ in f_steps if s_["iteration"] == 1] or [0.0]))
            its = [s_["iteration"] for s_ in f_steps if s_["alignment_ended"]]
            families[kind] = {"ms_per_step": round(f_ms, 4), "over_sheet": round(f_ms / sheet_ms, 3),
                              "first_launch_search_ms": round(f_first, 4),
                              "first_launch_over_sheet": round(f_first / sheet_first, 2) if sheet_first > 0 else None,
                              "iterations_per_alignment": its[0] if its else None,
                              "value": round(float(sum(s_["num_correspondences"] for s_ in f_steps)) / f_elapsed, 1)}
            del f_icp, f_ne, f_tree, f_tgt, f_src

    out = None
    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # an N=1 figure (it would stall the other ranks)
            cpu = cpu_baseline(args, mode, n, tgt_h, src_h)
        boundary = None
        if not args.no_host_align and world == 1:
            boundary = host_align(mode, tgt_h, src_h)
        name = {2: "config 2: 2^20-point clouds, k=1 NN + point-to-point ICP (SVD)",
                3: "config 3: 10M-point clouds, k=%d NormalEstimation + point-to-plane ICP" % args.knn}[cfg]
        out = {
            "metric": "ICP correspondences/sec", "value": round(ncorr / elapsed, 1), "unit": "correspondences/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 search / f64 accumulate", "data": "synthetic",
            "config": {"workload": "%s; %d-point synthetic %s source per GPU vs %d-point target, 1-NN "
                                   "correspondences, max_dist 0.1" % (name, n, {"sheet": "Gaussian-surface"}.get(args.cloud, args.cloud), n),
                       "cloud": args.cloud, "options": args.opt or None,
                       "baseline_metric": "ICP correspondences/sec/GPU + ms/iteration, 10M-pt cloud; HBM GB/s vs roofline "
                                          "(BASELINE.json; `value` is the whole-job aggregate, ms/iteration = ms_per_step, "
                                          "HBM GB/s = roofline.achieved)",
                       "points_per_gpu": n, "target_points": n, "mode": "p2plane" if mode == 1 else "p2point",
                       "rejectors": args.rejectors or None, "reciprocal": bool(args.reciprocal),
                       "loop": "device-driven (pclhip_icp_run_steps): search, accumulate, reduce, solve + convergence "
                               "kernels queued back to back",
                       "parallelism": "source slab sharded x%d, target replicated%s" %
                                      (world, (", ncclAllReduce of the 32-double record per iteration (%s)" %
                                               {"native": "issued from C on the context's stream",
                                                "torch": "torch.distributed on the context's stream"}[collective])
                                       if world > 1 else "")},
            "roofline": roofline, "cpu_baseline": cpu, "families": families,
            "per_step": [{"iteration": s["iteration"], "search_ms": round(s["search_ms"], 4),
                          "step_ms": round(s["step_ms"], 4), "ended": s["alignment_ended"], "state": s["state"]} for s in steps],
            "setup": {"index_build_ms": round(build_ms, 3), "index_build_first_ms": round(build_first_ms, 3),
                      "index_build_GBps_alg": round(B_ALG_BUILD * n / (build_ms * 1e-3) / 1e9, 1),
                      "index_build_roofline_frac": round(B_ALG_BUILD * n / (build_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                      "normals_kernel_ms": None if normals_ms is None else round(normals_ms, 3),
                      "normals_kernel_first_ms": None if normals_first_ms is None else round(normals_first_ms, 3),
                      "normals_GBps_alg": None if normals_ms is None else
                      round(B_ALG_NORMALS * n / (normals_ms * 1e-3) / 1e9, 1),
                      "normals_roofline_frac": None if normals_ms is None else
                      round(B_ALG_NORMALS * n / (normals_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                      "iteration_roofline_frac": round(B_ALG_ITER[mode] * corr_per_launch_local /
                                                       (kernel_ms / max(args.steps, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                      "source_order_ms": round(source_order_ms, 3),
                      # a new source against the resident target: ordering + one whole alignment (device times)
                      "fresh_frame_ms": None if fresh_frame is None else fresh_frame["fresh_frame_ms"], "fresh_frame": fresh_frame,
                      # pcl::Registration::align() from HOST pcl::PointClouds through the real-PCL binding (mock build)
                      "host_align_ms": None if not boundary or "align_first" not in boundary else
                      round(boundary["align_first"]["total_ms"], 3),
                      "host_align": boundary,
                      "synth_gen_s": round(gen_s, 1), "commit": git_head()},
        }
    finish(out, rank, world)


def run_pipeline(args, ctx, comm, tgt, src, n, rank, world, fence, gen_s):
    """config 4: VoxelGrid(0.01) on both clouds -> target index -> NormalEstimation(k) -> alignment.  One step =
    the whole pipeline on the resident 10M-point clouds."""
    import torch
    import torch.distributed as dist
    import pcl_amd

    stage_ms = {"voxelgrid": [], "index_build": [], "normals": [], "icp": [], "source_order": []}
    info = {}

    def step():
        t0 = time.perf_counter()
        filt = []
        for cloud in (tgt, src):
            vg = pcl_amd.VoxelGrid(ctx)
            vg.setInputCloud(cloud)
            vg.setLeafSize(0.01, 0.01, 0.01)
            filt.append(vg.filter())
        ctx.synchronize()
        t1 = time.perf_counter()
        tree = pcl_amd.KdTree(ctx)
        tree.setInputCloud(filt[0])
        ne = pcl_amd.NormalEstimation(ctx)
        ne.setInputCloud(filt[0])
        ne.setSearchMethod(tree)
        ne.setKSearch(args.knn)
        ne.setViewPoint(0, 0, 10)
        ne.compute(want_output=False)
        icp = pcl_amd.IterativeClosestPointWithNormals(ctx)
        icp.setSearchMethodTarget(tree, True)
        icp.setInputSource(filt[1])
        icp.setMaximumIterations(20)
        icp.setMaxCorrespondenceDistance(0.1)
        icp.setTransformationEpsilon(1e-10)
        from pcl_amd.dist import attach_collective
        attach_collective(icp, comm, torch.cuda.current_device(), world)
        icp.align()
        t2 = time.perf_counter()
        stage_ms["voxelgrid"].append((t1 - t0) * 1e3)
        stage_ms["index_build"].append(tree.build_ms())
        stage_ms["normals"].append(tree.lastKernelMs())
        stage_ms["icp"].append(float(icp.result.gpu_ms))
        stage_ms["source_order"].append(icp.sourceOrderMs())
        info.update(filtered=(int(filt[0].shape[0]), int(filt[1].shape[0])), iterations=icp.nr_iterations_,
                    T=icp.getFinalTransformation(), wall_rest_ms=(t2 - t1) * 1e3)

    for _ in range(args.warmup):
        step()
    for v in stage_ms.values():
        v.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    from pcl_amd.dist import max_over_ranks
    elapsed = max_over_ranks(time.perf_counter() - t0, world)
    if rank != 0:
        return None
    from pcl_amd import synth
    vg_ms = float(np.mean(stage_ms["voxelgrid"]))
    achieved = B_ALG_VOXEL * 2 * n / (vg_ms * 1e-3) / 1e9
    return {
        "metric": "VoxelGrid(0.01) + NormalEstimation + point-to-plane ICP pipeline, input points/sec",
        "value": round(2.0 * n * args.steps * world / elapsed, 1), "unit": "input points/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 / f64 accumulate",
        "data": "synthetic",
        "config": {"workload": "config 4: two %d-point clouds -> VoxelGrid(0.01) -> %d / %d points -> k=%d normals -> "
                               "point-to-plane ICP (%d iterations), everything on the device" %
                               (n, info["filtered"][0], info["filtered"][1], args.knn, info["iterations"]),
                   "points_per_gpu": n, "parallelism": "replicas x%d" % world},
        "roofline": {"bound": "hbm", "kernel": "vg_* (VoxelGrid of both clouds, wall time of the two filter() calls)",
                     "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "alg_bytes_per_input_point": B_ALG_VOXEL},
        "cpu_baseline": None if args.no_cpu_baseline or world > 1 else cpu_pipeline(args, tgt, src),
        "stages_ms": {k: round(float(np.mean(v)), 3) for k, v in stage_ms.items()},
        "result": {"T_minus_T_gt_frobenius": float(np.linalg.norm(info["T"] - synth.ground_truth_transform()))},
        "setup": {"synth_gen_s": round(gen_s, 1), "commit": git_head()},
    }


def host_align(mode, tgt_h, src_h):
    """The boundary a PCL user sees: examples/bench_pcl_align.cpp drives pcl::Registration::align() on
    IterativeClosestPoint[WithNormals]HIP with HOST pcl::PointClouds of the bench's own clouds (compiled against the PCL
    mock: PCL's classes with their real signatures).  Returns its JSON (milliseconds per stage) or a note on failure."""
    import tempfile
    try:
        d = tempfile.mkdtemp(prefix="pclhip_align_")
        exe = os.path.join(d, "bench_pcl_align")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "tests", "cpp", "pcl_mock"),
                               os.path.join(ROOT, "examples", "bench_pcl_align.cpp"), "-o", exe,
                               "-L" + os.path.join(ROOT, "pcl_amd"), "-lpclhip",
                               "-Wl,-rpath," + os.path.join(ROOT, "pcl_amd")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        tgt_h.astype(np.float32).tofile(os.path.join(d, "t.f32"))
        src_h.astype(np.float32).tofile(os.path.join(d, "s.f32"))
        out = subprocess.run([exe, os.path.join(d, "t.f32"), os.path.join(d, "s.f32"), str(len(tgt_h)), str(mode)],
                             capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        res = json.loads(line[-1]) if line else {"error": (out.stderr or out.stdout)[-300:]}
        for f in ("t.f32", "s.f32", "bench_pcl_align"):
            try:
                os.remove(os.path.join(d, f))
            except OSError:
                pass
        return res
    except Exception as e:  # no compiler on the box, ...: the bench line does not depend on it
        return {"error": repr(e)[:300]}


def host_cpus():
    """(hardware threads, physical cores) of this box: cpu_baseline.cores is the number of THREADS the oracle ran on."""
    threads = os.cpu_count() or 1
    cores = set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return threads, (len(cores) or None)


def cpu_quota():
    """CPU time this container may use, in cores (cgroup v2 cpu.max, v1 cfs quota), or None when unlimited / unknown.  A GPU box
    shows every hardware thread of its host to a container that may only run on a few cores' worth of time: threads
    beyond the quota are throttled, not run."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return float(q) / float(per)
        return None
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / per if q > 0 else None
    except (OSError, ValueError):
        return None


def cpu_pipeline(args, tgt, src):
    """config 4 on the host cores: the oracle's VoxelGrid (single thread, as in PCL) + kd-tree + normals + ICP."""
    from oracle import pcl_oracle as orc
    cores = orc.default_threads()
    tgt_h, src_h = tgt.cpu().numpy(), src.cpu().numpy()
    t0 = time.perf_counter()
    ft = orc.voxelgrid(tgt_h, 0.01)[0]
    fs = orc.voxelgrid(src_h, 0.01)[0]
    t1 = time.perf_counter()
    tree = orc.KdTree(ft)
    nrm, _ = tree.normals(ft, args.knn, viewpoint=(0, 0, 10), nthreads=cores)
    r = orc.icp_align(tree, ft, fs, mode=1, tgt_normals=nrm, max_iterations=20, nthreads=cores,
                      max_correspondence_distance=0.1, transformation_epsilon=1e-10)
    t2 = time.perf_counter()
    return {"value": round(2.0 * len(tgt_h) / (t2 - t0), 1), "unit": "input points/s", "cores": cores,
            "threads_used": cores, "host_hardware_threads": host_cpus()[0], "host_physical_cores": host_cpus()[1],
            "kind": "port",
            "note": "a dependency-free RESTATEMENT of PCL's CPU path (oracle/pcl_oracle.c), not PCL + FLANN: a stated baseline, "
                    "never a target -- its one-thread search costs several microseconds per query where FLANN-class trees take "
                    "1-2, so the GPU / CPU ratio says nothing about kernel quality (roofline.frac does)",
            "sample": "the same two %d-point clouds, whole pipeline once: VoxelGrid %.2f s (1 thread, as in PCL), "
                      "kd-tree + k=%d normals + %d ICP iterations %.2f s (search on %d threads)" %
                      (len(tgt_h), t1 - t0, args.knn, r["iterations"], t2 - t1, cores)}


def morton_order(pts):
    """Permutation that puts a cloud into Morton order of its bounding box (21 bits per axis): spatially ordered queries, the
    order a scan or any spatial pre-sort gives a source cloud.  numpy, host; not timed."""
    import numpy as np
    p = np.asarray(pts[:, :3], np.float64)
    lo, hi = p.min(axis=0), p.max(axis=0)
    q = ((p - lo) / np.maximum(hi - lo, 1e-30) * ((1 << 21) - 1)).astype(np.uint64)

    def spread(v):
        v = v & np.uint64(0x1FFFFF)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v
    return np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2)), kind="stable")


def interleave_memory(on):
    """set_mempolicy(MPOL_INTERLEAVE over all NUMA nodes) for the pages this process touches from now on (the oracle's
    tree is built by one thread: first-touch would put all of it on that thread's node, and 256 threads would then query
    one node's memory).  Raw syscall (no libnuma needed); returns whether it took effect."""
    try:
        import ctypes
        nodes = [int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()]
        if len(nodes) < 2:
            return False
        libc = ctypes.CDLL(None, use_errno=True)
        maxnode = max(nodes) + 2
        mask = (ctypes.c_ulong * ((maxnode + 63) // 64))()
        if on:
            for nd in nodes:
                mask[nd // 64] |= 1 << (nd % 64)
        SYS_set_mempolicy, MPOL_DEFAULT, MPOL_INTERLEAVE = 238, 0, 3   # x86_64
        rc = libc.syscall(SYS_set_mempolicy, MPOL_INTERLEAVE if on else MPOL_DEFAULT, mask if on else None,
                          maxnode if on else 0)
        return rc == 0
    except Exception:
        return False


def cpu_baseline(args, mode, n, tgt, src):
    """The oracle (restated PCL KdTree + ICP, OpenMP over source points as in correspondence_estimation.hpp
    :163-191, estimation serial as in PCL) on the SAME clouds as the GPU line, timed on this box's host cores at
    T = 1 (bounded slice), T = physical cores and T = hardware threads.  The CPU gets the conditions a careful user would
    give it: the source in a spatial (Morton) order, so that a thread's contiguous share of the loop stays inside one part of
    the tree, and the tree's pages interleaved over the NUMA nodes; the figure with the source in its given (random) order
    is reported next to it."""
    import numpy as np
    from oracle import pcl_oracle as orc
    hw_threads, phys = host_cpus()
    quota = cpu_quota()
    # threads of the headline figure: every hardware thread -- unless the container's CPU quota is smaller, in which case more
    # threads than the quota only add throttling (measured on a 256-thread host with a 16-core quota: 256 threads reach
    # 11x one thread, the quota's worth of threads is what the box can give)
    cores = orc.default_threads()
    if quota is not None and quota < cores:
        cores = max(1, int(round(quota)))
    interleaved = interleave_memory(True)
    t0 = time.perf_counter()
    tree = orc.KdTree(tgt)
    build_s = time.perf_counter() - t0
    nrm = None
    normals_s = None
    if mode == 1:
        t0 = time.perf_counter()
        nrm, _ = tree.normals(tgt, args.knn, viewpoint=(0, 0, 10), nthreads=cores)
        normals_s = time.perf_counter() - t0
    src_sorted = np.ascontiguousarray(src[morton_order(src)])
    if interleaved:
        interleave_memory(False)
    kw = dict(mode=mode, tgt_normals=nrm, max_correspondence_distance=0.1, transformation_epsilon=0.0)
    # one thread first, on a bounded slice: its per-query cost also sizes the all-threads samples so that the baseline stays
    # within ~30 s of CPU wall time on a host with few cores (on the 128-core boxes seen so far the sample is the whole cloud)
    m1 = min(n, 1_000_000)
    r1 = orc.icp_align(tree, tgt, src_sorted[:m1], max_iterations=1, nthreads=1, **kw)
    per_iter1 = r1["seconds_total"] / max(r1["iterations"], 1)
    est_full_iter = per_iter1 / m1 * n / max(cores * 0.25, 1.0)
    n_all = n if est_full_iter * 3 <= 20.0 else max(m1, int(n * 20.0 / (est_full_iter * 3)))

    def run(cloud, threads, iters):
        r_ = orc.icp_align(tree, tgt, cloud if n_all == n else cloud[:n_all], max_iterations=iters, nthreads=threads, **kw)
        it_ = max(r_["iterations"], 1)
        return r_, it_, r_["seconds_total"] / it_

    r, it, per_iter = run(src_sorted, cores, 3)
    n_sample = n_all
    out = {"value": round(r["num_correspondences"] / per_iter, 1), "unit": "correspondences/s", "cores": cores,
           "threads_used": cores, "host_hardware_threads": hw_threads, "host_physical_cores": phys,
           "container_cpu_quota_cores": quota,
           "kind": "port",
           "sample": "the bench's own %d-point target and %s%d-point source in Morton order, %d ICP iterations on %d threads "
                     "(search %.3f s + serial estimate/transform %.3f s per iteration); 1 thread: one iteration over "
                     "the first %d source points against the full target; kd-tree build %.2f s single-thread (as in "
                     "FLANN)%s; tree pages %s" %
                     (n, "" if n_sample == n else "the first %d points of its " % n_sample, n, it, cores, r["seconds_search"] / it,
                      (r["seconds_total"] - r["seconds_search"]) / it, m1, build_s,
                      "" if normals_s is None else ", k=%d normals %.2f s on %d threads" % (args.knn, normals_s, cores),
                      "interleaved over the NUMA nodes" if interleaved else "first-touch (one NUMA node, or the policy call was refused)"),
           "ms_per_iteration": round(per_iter * 1e3, 2),
           "search_ms_per_iteration": round(r["seconds_search"] / it * 1e3, 2),
           "serial_ms_per_iteration": round((r["seconds_total"] - r["seconds_search"]) / it * 1e3, 2),
           "single_thread": {"value": round(r1["num_correspondences"] / per_iter1, 1), "unit": "correspondences/s",
                             "cores": 1, "us_per_query": round(r1["seconds_search"] / m1 * 1e6, 3),
                             "sample_points": m1}}
    one = r1["seconds_search"] / m1
    out["search_speedup_over_one_thread"] = round(one / (r["seconds_search"] / it / n_sample), 1)
    if quota is not None and quota < hw_threads:   # every hardware thread all the same: what oversubscribing the quota gives
        ra, ita, per_a = run(src_sorted, hw_threads, 2)
        out["all_hardware_threads"] = {"value": round(ra["num_correspondences"] / per_a, 1), "unit": "correspondences/s",
                                       "cores": hw_threads, "search_ms_per_iteration": round(ra["seconds_search"] / ita * 1e3, 2),
                                       "search_speedup_over_one_thread": round(one / (ra["seconds_search"] / ita / n_sample), 1)}
    if phys and phys < cores:   # one thread per physical core
        rp, itp, per_p = run(src_sorted, phys, 2)
        out["physical_cores"] = {"value": round(rp["num_correspondences"] / per_p, 1), "unit": "correspondences/s", "cores": phys,
                                 "search_ms_per_iteration": round(rp["seconds_search"] / itp * 1e3, 2),
                                 "search_speedup_over_one_thread": round(one / (rp["seconds_search"] / itp / n_sample), 1)}
    ru, itu, per_u = run(src, cores, 2)   # the source as the bench generated it: random order
    out["unsorted_source"] = {"value": round(ru["num_correspondences"] / per_u, 1), "unit": "correspondences/s", "cores": cores,
                              "search_ms_per_iteration": round(ru["seconds_search"] / itu * 1e3, 2),
                              "search_speedup_over_one_thread": round(one / (ru["seconds_search"] / itu / n_sample), 1)}
    return out


def finish(out, rank, world):
    import torch.distributed as dist
    if rank == 0 and out is not None:
        # RCCL (NCCL_DEBUG=VERSION on this image) writes its banner through C stdio, which would otherwise be
        # flushed after this line at exit: push it out first so the JSON line is the last thing on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
