"""One rank of a world_size-N run of the library's own multi-GPU code on the CPU emulation (tests/test_wavesim.py starts N
of these as processes; never used outside the tests).  argv: mode rank world workdir [points]

  mode "target"  the target cut into kd slabs + halo (pcl_amd.dist.ShardedTarget), the whole source on every rank, every
                 rank serving the source points whose current position lies in its region (bench.py --config 5)
  mode "source"  the whole target indexed on every rank, the source cut into contiguous slabs (bench.py --gpus N)

Either way the ranks share a native communicator (pclhip_comm_*: the id travels through a file here, through
torch.distributed in bench.py) and the device-driven loop all-reduces the 32-double record of every iteration between
its reduction and its solve kernel.  Every rank writes what it saw to workdir/rank<r>.npz.
"""
import os
import sys
import time

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
uid_path = os.path.join(work, "uid.bin")
if rank == 0:
    uid = pcl_amd.Communicator.unique_id()
    with open(uid_path + ".tmp", "wb") as f:
        f.write(bytes(uid))
    os.rename(uid_path + ".tmp", uid_path)
else:
    t0 = time.time()
    while not os.path.exists(uid_path):
        if time.time() - t0 > 120:
            raise SystemExit("rank 0 never published the communicator id")
        time.sleep(0.05)
    uid = open(uid_path, "rb").read()
ctx = pcl_amd.Context(0)
comm = pcl_amd.Communicator(ctx, rank, world, bytes(uid))
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
steps = icp.runSteps(6)
served = len(icp.fetchCorrespondences()[0])
fit = icp.getFitnessScore(0.01)
np.savez(os.path.join(work, "rank%d.npz" % rank), T=T, iterations=iters, converged=icp.hasConverged(),
         counts=np.asarray([s["num_correspondences"] for s in steps], np.float64),
         step_iterations=np.asarray([s["iteration"] for s in steps]), served=served, index_points=tree.size(),
         fitness=fit, fitness_points=icp.fitness_points, kept_after_align=kept_after_align)
