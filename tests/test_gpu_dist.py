"""The RCCL leg of the multi-GPU path on ONE GPU: a world_size-1 "nccl" group exercises the real hook
(device record wrapped without a copy, all-reduce on the context's stream, host read after it).
The 8-GPU run itself is the driver's; the N>1 arithmetic is covered on CPU in test_dist_gloo.py."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_allreduce_hook_over_rccl_single_rank():
    import torch
    import torch.distributed as dist
    import pcl_amd
    from pcl_amd import synth
    from pcl_amd.dist import device_doubles, make_allreduce_hook
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        # the zero-copy wrapper really aliases device memory
        buf = torch.arange(32, dtype=torch.float64, device="cuda")
        alias = device_doubles(buf.data_ptr(), 32, 0)
        alias += 1
        assert torch.equal(buf.cpu(), torch.arange(32, dtype=torch.float64) + 1)
        ctx = pcl_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        tgt = synth.gaussian_surface(200_000, synth.TARGET_SEED)
        src = synth.apply_rigid(np.linalg.inv(synth.ground_truth_transform()),
                                synth.gaussian_surface(100_000, synth.SOURCE_SEED))
        tree = pcl_amd.KdTree(ctx)
        tree.setInputCloud(tgt)
        ne = pcl_amd.NormalEstimation(ctx)
        ne.setInputCloud(tgt)
        ne.setSearchMethod(tree)
        ne.setKSearch(8)
        ne.compute(want_output=False)
        results = []
        for use_hook in (False, True):
            icp = pcl_amd.IterativeClosestPointWithNormals(ctx)
            icp.setSearchMethodTarget(tree)
            icp.setInputSource(src)
            icp.setMaximumIterations(6)
            icp.setMaxCorrespondenceDistance(0.1)
            if use_hook:
                icp.setAllReduce(make_allreduce_hook(0))
            icp.align()
            sums = icp.iterate(np.eye(4, dtype=np.float32), max_dist=0.1)
            results.append((icp.getFinalTransformation().copy(), icp.nr_iterations_, sums.copy()))
        # a 1-rank sum is the identity: bit-identical results with and without the collective
        assert np.array_equal(results[0][0], results[1][0]) and results[0][1] == results[1][1]
        assert np.array_equal(results[0][2], results[1][2])
    finally:
        dist.destroy_process_group()
