"""One rank of a world_size-N run of the library's own multi-GPU code on the CPU emulation (tests/test_wavesim.py starts N
of these as processes; never used outside the tests).  argv: mode rank world workdir [points]

  mode "target"  the target cut into kd slabs + halo (pcl_amd.dist.ShardedTarget), the whole source on every rank, every
                 rank serving the source points whose current position lies in its region (bench.py --config 5)
  mode "source"  the whole target indexed on every rank, the source cut into contiguous slabs (bench.py --gpus N)

Either way the ranks share a native communicator (pclhip_comm_*), brought up by the very functions bench.py uses
(pcl_amd/dist.py: init_ranks / native_communicator / make_fence / timed_steps; the torch group is gloo here, nccl there) and the device-driven loop all-reduces the 32-double record of every iteration between
its reduction and its solve kernel.  Every rank writes what it saw to workdir/rank<r>.npz.
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pcl_amd  # noqa: E402
from pcl_amd import synth  # noqa: E402
from pcl_amd.dist import ShardedTarget, shard_range  # noqa: E402

mode, rank, world, work = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
mode, _, extra = mode.partition("+")   # "+rej": MedianDistance + Trimmed + Distance chain; "+recip": reciprocal correspondences
extras = set(extra.split("+")) if extra else set()   # "+empty" (source mode): the LAST rank's share of the source is empty
n = int(sys.argv[5]) if len(sys.argv) > 5 else 60_000
# the bring-up of bench.py, function for function (pcl_amd/dist.py): rendezvous from the launcher's environment (gloo here, nccl
# there), rank 0's communicator id broadcast over the torch group, the native communicator next to torch's, the fences
os.environ["RANK"], os.environ["LOCAL_RANK"], os.environ["WORLD_SIZE"] = str(rank), "0", str(world)
from pcl_amd.dist import init_ranks, make_fence, native_communicator, timed_steps  # noqa: E402
rank, _, world = init_ranks("gloo")
ctx = pcl_amd.Context(0)
fence = make_fence(ctx, world)
comm = native_communicator(ctx, rank, world)
tgt, src, _ = synth.icp_pair(n)
region = None
if mode == "target":
    st = ShardedTarget(ctx, tgt, rank, world, 0.1, k_normals=8, viewpoint=(0, 0, 10))
    tree, region = st.tree, st.region
    my_src = src
else:
    tree = pcl_amd.KdTree(ctx)
    tree.setInputCloud(tgt)
    ne = pcl_amd.NormalEstimation(ctx)
    ne.setInputCloud(tgt)
    ne.setSearchMethod(tree)
    ne.setKSearch(8)
    ne.setViewPoint(0, 0, 10)
    ne.compute(want_output=False)
    if "empty" in extras:
        start, count = shard_range(n, rank, world - 1) if rank < world - 1 else (0, 0)
    else:
        start, count = shard_range(n, rank, world)
    my_src = np.ascontiguousarray(src[start:start + count])
icp = pcl_amd.IterativeClosestPointWithNormals(ctx)
icp.setSearchMethodTarget(tree, True)
icp.setInputSource(my_src)
icp.setMaximumIterations(20)
icp.setMaxCorrespondenceDistance(0.1)
icp.setTransformationEpsilon(1e-10)
icp.setCommunicator(comm)
if region is not None:
    icp.setRegion(region)
if "rej" in extras:
    a = pcl_amd.CorrespondenceRejectorMedianDistance()
    a.setMedianFactor(1.5)
    b = pcl_amd.CorrespondenceRejectorTrimmed()
    b.setOverlapRatio(0.8)
    d = pcl_amd.CorrespondenceRejectorDistance()
    d.setMaximumDistance(0.05)
    for r in (a, b, d):
        icp.addCorrespondenceRejector(r)
if "o2o" in extras:   # OneToOne under TARGET sharding: the per-target minimum keys are reduced over the ranks
    icp.addCorrespondenceRejector(pcl_amd.CorrespondenceRejectorOneToOne())
if "recip" in extras:
    icp.setUseReciprocalCorrespondences(True)
icp.align()
T = icp.getFinalTransformation().copy()
iters = icp.nr_iterations_
kept_after_align = len(icp.fetchCorrespondences()[0])   # the pairs of the last iteration that this rank serves and the chain kept
# the measurement loop of bench.py on top: whole alignments queued back to back, records all-reduced
steps, elapsed = timed_steps(icp, 6, 0, fence, world)   # warm-up 0: the records below are those of the first six steps
assert elapsed > 0.0
served = len(icp.fetchCorrespondences()[0])
fit = icp.getFitnessScore(0.01)
np.savez(os.path.join(work, "rank%d.npz" % rank), T=T, iterations=iters, converged=icp.hasConverged(),
         counts=np.asarray([s["num_correspondences"] for s in steps], np.float64),
         step_iterations=np.asarray([s["iteration"] for s in steps]), served=served, index_points=tree.size(),
         fitness=fit, fitness_points=icp.fitness_points, kept_after_align=kept_after_align)
import torch.distributed as dist  # noqa: E402
dist.barrier()
dist.destroy_process_group()
