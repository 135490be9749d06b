#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: ICP correspondences/s (+ ms/iteration) on the
10M-point synthetic Gaussian-surface cloud, k=8 normals + point-to-plane ICP, on N MI355X GPUs.

A "step" is ONE ICP iteration over the whole (rank-local) source cloud: transform ->
exact 1-NN in the target -> 6x6 normal-system accumulation -> (all-reduce) -> host solve.
Iterations are drawn from repeated IterativeClosestPointWithNormals::align() runs on the config's
clouds (point-to-plane converges in ~3-4 iterations): when an alignment
converges the working cloud is rewound and the next alignment starts, so the K timed steps contain
the realistic mix of far-from-aligned and nearly-aligned iterations.  Target index + normals are
built before the timed region (reported separately) and stay resident in HBM.

Multi-GPU (weak scaling): every rank holds the full target index (it fits HBM many times over;
north_star shards the target only when it does not) and its own slab of the source
(n_points source points per rank, disjoint counter ranges of the same surface); the only exchange
per iteration is the all-reduce of the 32-double reduction record over RCCL.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic bytes per correspondence: src 16 + tgt 16 + (idx 4 + d2 4) = 40 for the
# search; + normal 16 = 56 for a whole point-to-plane iteration
B_ALG_SEARCH = 40.0
B_ALG_ITER = {1: 56.0, 0: 40.0}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--points", type=int, default=10_000_000, help="source points per GPU (= target points)")
    ap.add_argument("--mode", choices=["p2plane", "p2point"], default="p2plane")
    ap.add_argument("--knn", type=int, default=8, help="k of NormalEstimation")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="points of the CPU-baseline sample")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    import pcl_amd
    from pcl_amd import synth

    mode = 1 if args.mode == "p2plane" else 0
    n = args.points
    stream = torch.cuda.current_stream().cuda_stream
    ctx = pcl_amd.Context(local_rank, stream=stream)

    # ---- synthetic clouds; target identical on every rank, source = this rank's slab
    t0 = time.perf_counter()
    tgt_h = synth.gaussian_surface(n, synth.TARGET_SEED)
    src_h = synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()),
                              synth.gaussian_surface(n, synth.SOURCE_SEED, start=rank * n))
    gen_s = time.perf_counter() - t0
    tgt = torch.from_numpy(tgt_h).cuda()
    src = torch.from_numpy(src_h).cuda()

    # ---- target index + normals (one-off, outside the timed region)
    tree = pcl_amd.KdTree(ctx)
    tree.setInputCloud(tgt)
    build_ms = tree.build_ms()
    normals_ms = None
    if mode == 1:
        ne = pcl_amd.NormalEstimation(ctx)
        ne.setInputCloud(tgt)
        ne.setSearchMethod(tree)
        ne.setKSearch(args.knn)
        ne.setViewPoint(0, 0, 10)
        ne.compute(want_output=False)
        normals_ms = tree.lastKernelMs()

    cls = pcl_amd.IterativeClosestPointWithNormals if mode == 1 else pcl_amd.IterativeClosestPoint
    icp = cls(ctx)
    icp.setSearchMethodTarget(tree)
    icp.setInputSource(src)
    max_dist = 0.1
    if world > 1:
        from pcl_amd.dist import make_allreduce_hook
        icp.setAllReduce(make_allreduce_hook(local_rank))  # RCCL all-reduce of the 32-double record

    state = {"T": np.eye(4, dtype=np.float32), "it": 0}
    icp.reset()

    def step():
        sums = icp.iterate(state["T"], max_dist=max_dist)
        T = icp.solve(sums)
        state["it"] += 1
        # DefaultConvergenceCriteria TRANSFORM test with transformation_epsilon 1e-10 (+ iteration cap 20)
        cos_angle = 0.5 * (float(T[0, 0]) + float(T[1, 1]) + float(T[2, 2]) - 1.0)
        tr2 = float(T[0, 3]) ** 2 + float(T[1, 3]) ** 2 + float(T[2, 3]) ** 2
        if (cos_angle >= 0.99999 and tr2 <= 1e-10) or state["it"] >= 20 or sums[28] < 3:
            icp.reset()                      # next alignment starts from the input cloud
            state["T"] = np.eye(4, dtype=np.float32)
            state["it"] = 0
        else:
            state["T"] = T
        return sums[28], icp.lastKernelMs(), icp.lastSearchMs()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    ncorr = 0.0
    kernel_ms = 0.0
    search_ms = 0.0
    for _ in range(args.steps):
        c, kms, sms = step()
        ncorr += c            # already the all-reduced (global) count when world > 1
        kernel_ms += kms
        search_ms += sms
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (the exact 1-NN search), live HIP-event timing on the context's
    # stream.  The iteration is two kernels: icp_search_kernel (dominant) and the streaming
    # icp_accumulate_kernel; with PCLHIP_ICP_FUSED=1 both run as one kernel and the two times coincide.
    fused = os.environ.get("PCLHIP_ICP_FUSED", "0") == "1"
    b_alg = B_ALG_ITER[mode] if fused else B_ALG_SEARCH
    avg_kernel_s = search_ms / args.steps / 1e3
    corr_per_launch_local = ncorr / args.steps / world
    achieved = b_alg * corr_per_launch_local / avg_kernel_s / 1e9
    roofline = {"bound": "hbm", "kernel": ("icp_iterate_kernel<%d>" % mode) if fused else "icp_search_kernel",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": None, "alg_bytes_per_corr": b_alg, "avg_kernel_ms": round(avg_kernel_s * 1e3, 4),
                "iteration_kernels_ms": round(kernel_ms / args.steps, 4),
                "iteration_alg_bytes_per_corr": B_ALG_ITER[mode]}
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tp):  # HBM bytes per launch from the committed PMC passes (profiles/README.md)
        try:
            roofline["traffic"] = json.load(open(tp)).get(
                ("icp_iterate_bytes_per_launch_%s" % args.mode) if fused else "icp_search_bytes_per_launch")
        except Exception:
            pass

    out = None
    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is an N=1 figure (it would stall the other ranks)
            cpu = cpu_baseline(args, mode)
        out = {
            "metric": "ICP correspondences/sec", "value": round(ncorr / elapsed, 1), "unit": "correspondences/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 search / f64 accumulate", "data": "synthetic",
            "config": {"workload": "%dM-point synthetic Gaussian-surface cloud per GPU, k=%d NormalEstimation + "
                                   "%s ICP, 1-NN correspondences, max_dist 0.1" %
                                   (n // 1_000_000, args.knn, "point-to-plane" if mode == 1 else "point-to-point"),
                       "baseline_metric": "ICP correspondences/sec/GPU + ms/iteration, 10M-pt cloud; HBM GB/s vs roofline "
                                          "(BASELINE.json; `value` is the whole-job aggregate, ms/iteration = ms_per_step, "
                                          "HBM GB/s = roofline.achieved)",
                       "points_per_gpu": n, "target_points": n, "mode": args.mode,
                       "parallelism": "source slab sharded x%d (kd-ordered per rank), target replicated" % world},
            "roofline": roofline, "cpu_baseline": cpu,
            "setup": {"index_build_ms": round(build_ms, 3),
                      "normals_kernel_ms": None if normals_ms is None else round(normals_ms, 3),
                      "synth_gen_s": round(gen_s, 1)},
        }
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


def cpu_baseline(args, mode):
    """The oracle (restated PCL KdTree+ICP, OpenMP over source points as in correspondence_estimation.hpp
    :163-191) on a bounded sample of the same workload, timed on this box's host cores."""
    from oracle import pcl_oracle as orc
    from pcl_amd import synth
    m = min(args.cpu_sample, args.points)
    tgt = synth.gaussian_surface(m, synth.TARGET_SEED)
    src = synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()), synth.gaussian_surface(m, synth.SOURCE_SEED))
    cores = orc.default_threads()
    t0 = time.perf_counter()
    tree = orc.KdTree(tgt)
    build_s = time.perf_counter() - t0
    nrm = None
    normals_s = None
    if mode == 1:
        t0 = time.perf_counter()
        nrm, _ = tree.normals(tgt, args.knn, viewpoint=(0, 0, 10), nthreads=cores)
        normals_s = time.perf_counter() - t0
    iters = 3
    r = orc.icp_align(tree, tgt, src, mode=mode, tgt_normals=nrm, max_iterations=iters, nthreads=cores,
                      max_correspondence_distance=0.1, transformation_epsilon=0.0)
    per_iter = r["seconds_total"] / max(r["iterations"], 1)
    return {"value": round(r["num_correspondences"] / per_iter, 1), "unit": "correspondences/s", "cores": cores,
            "kind": "port",
            "sample": "%d-point target + %d-point source of the same surface, %d ICP iterations (search OpenMP over "
                      "%d threads, estimation serial as in PCL); kd-tree build %.2f s single-thread%s" %
                      (m, m, r["iterations"], cores, build_s,
                       "" if normals_s is None else ", k=%d normals %.2f s" % (args.knn, normals_s)),
            "ms_per_iteration": round(per_iter * 1e3, 2)}


if __name__ == "__main__":
    main()
