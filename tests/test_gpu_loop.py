"""The device-driven ICP loop (pcl_amd/csrc/icp_loop.hip): iterations closed on the GPU by icp_solve_kernel
(solve + final = T * final + DefaultConvergenceCriteria, impl/icp.hpp:204-238,
impl/default_convergence_criteria.hpp:49-140) against its host twin and against the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu():
    import pcl_amd
    from conftest import make_context
    return make_context(0)


def _make_icp(gpu, tgt, src, mode, normals=None):
    import pcl_amd
    cls = pcl_amd.IterativeClosestPointWithNormals if mode == 1 else pcl_amd.IterativeClosestPoint
    icp = cls(gpu)
    icp.setInputTarget(tgt)
    if normals is not None:
        icp.setTargetNormals(normals)
    icp.setInputSource(src)
    icp.setMaximumIterations(20)
    icp.setMaxCorrespondenceDistance(0.1)
    icp.setTransformationEpsilon(1e-10)
    return icp


@pytest.mark.parametrize("mode", [0, 1])
def test_run_steps_is_align_repeated(gpu, mode):
    import pcl_amd
    from oracle import pcl_oracle as orc
    tgt, src, T_gt = pcl_amd.synth.icp_pair(100_000)
    normals = orc.KdTree(tgt).normals(tgt, 8, viewpoint=(0, 0, 10))[0] if mode == 1 else None
    icp = _make_icp(gpu, tgt, src, mode, normals)
    icp.align()
    k = icp.nr_iterations_
    T = icp.getFinalTransformation()
    state = icp.getConvergenceState()
    assert k >= 3
    # the criteria keep their memory across align() calls (as in the reference): a second align() on the same
    # object is the reference for the SECOND alignment of a stream, which starts like a fresh object
    icp.align()
    k2, T2, state2 = icp.nr_iterations_, icp.getFinalTransformation(), icp.getConvergenceState()
    steps = icp.runSteps(k + k2 + 2)
    assert len(steps) == k + k2 + 2
    assert [s["iteration"] for s in steps[:k]] == list(range(1, k + 1))
    assert [s["alignment_ended"] for s in steps[:k]] == [False] * (k - 1) + [True]
    assert steps[k - 1]["state"] == state and steps[k - 1]["converged"]
    assert np.array_equal(steps[k - 1]["final_transformation"], T)          # same kernels, same bits
    assert steps[k]["iteration"] == 1 and not steps[k]["alignment_ended"]   # the next alignment started
    assert [s["iteration"] for s in steps[k:k + k2]] == list(range(1, k2 + 1))
    assert steps[k + k2 - 1]["state"] == state2 and np.array_equal(steps[k + k2 - 1]["final_transformation"], T2)
    assert all(s["num_correspondences"] == len(src) for s in steps)
    assert all(s["search_ms"] > 0 and s["step_ms"] >= s["kernels_ms"] >= s["search_ms"] for s in steps)
    assert np.abs(T - T2).max() < 1e-5 and state != "NOT_CONVERGED" and abs(k2 - k) <= 1
    # a call boundary that cuts an alignment short must not leak into the next stream: every stream starts alike
    a = icp.runSteps(k + 1)
    b = icp.runSteps(k + 1)
    assert [(s["iteration"], s["state"]) for s in a] == [(s["iteration"], s["state"]) for s in b]
    assert [s["iteration"] for s in a] == list(range(1, k + 1)) + [1]
    # ... and the align() memory is untouched by the streams: a third align() continues from the second
    icp.align()
    assert icp.nr_iterations_ == k2 and np.array_equal(icp.getFinalTransformation(), T2)


def _host_loop_align(icp, conv):
    """IterativeClosestPoint::computeTransformation (impl/icp.hpp:113-268) driven from the HOST, composed from the C ABI's
    pieces: pclhip_icp_iterate (search + accumulate, record read back) -> pclhip_solve_transformation ->
    pclhip_convergence_has_converged.  The twin of the device-driven loop inside pclhip_icp_align (same kernels for the
    search and the sums, the solve and the criteria on the host instead of in icp_finalize_kernel)."""
    import ctypes as C
    from pcl_amd import _lib
    lib = _lib.load()
    icp.reset()
    final = np.eye(4, dtype=np.float32)
    T_apply = np.eye(4, dtype=np.float32)
    Tk = np.eye(4, dtype=np.float32)
    it, converged = 0, False
    while True:
        sums = icp.iterate(T_apply)
        ncorr = sums[28]
        if ncorr < icp.p.min_number_correspondences:          # icp.hpp:204-213
            conv.convergence_state = 5                          # NO_CORRESPONDENCES
            break
        Tk = icp.solve(sums)                                    # :216-217
        T_apply = Tk                                            # :220 (applied by the next launch)
        final = (Tk @ final).astype(np.float32)                 # :223 (float product, as the library's mat4_mul_f32)
        it += 1
        mse = sums[27] / ncorr
        t = np.ascontiguousarray(Tk, np.float32).reshape(16)
        converged = bool(lib.pclhip_convergence_has_converged(C.byref(icp.p), C.byref(conv), it,
                                                              t.ctypes.data_as(C.POINTER(C.c_float)), float(mse)))
        if converged or conv.convergence_state != 0:
            break
    return {"T": final, "it": it, "state": int(conv.convergence_state), "conv": converged, "last": Tk}


def test_device_loop_matches_host_loop_twin(gpu):
    # The device-driven loop against a host-driven one composed from pclhip_icp_iterate / pclhip_solve_transformation /
    # pclhip_convergence_has_converged (round 1's loop, which left the library in round 4: it is what a caller with a foreign
    # estimation plugged in writes).  Device and host libm may differ in the last ulp of a double sin/cos: the 4x4 agree to
    # float rounding, iteration counts and states exactly.
    import ctypes as C
    import pcl_amd
    from pcl_amd import _lib
    from oracle import pcl_oracle as orc
    z = np.load(os.path.join(ROOT, "tests", "golden", "bunny.npz"))

    def xyz1(a):
        o = np.ones((len(a), 4), np.float32)
        o[:, :3] = a[:, :3]
        return o
    states = ["NOT_CONVERGED", "ITERATIONS", "TRANSFORM", "ABS_MSE", "REL_MSE", "NO_CORRESPONDENCES", "FAILURE_AFTER_MAX_ITERATIONS"]
    cases = {"bunny": (xyz1(z["bun4"]), xyz1(z["bun0"]), 0, None, 0.05)}
    tgt, src, _ = pcl_amd.synth.icp_pair(60_000)
    cases["p2plane"] = (tgt, src, 1, orc.KdTree(tgt).normals(tgt, 8, viewpoint=(0, 0, 10))[0], 0.1)
    cases["p2point"] = (tgt, src, 0, None, 0.1)
    for name, (tgt, src, mode, nrm, md) in cases.items():
        def make():
            cls = pcl_amd.IterativeClosestPointWithNormals if mode == 1 else pcl_amd.IterativeClosestPoint
            icp = cls(gpu)
            icp.setInputTarget(tgt)
            if nrm is not None:
                icp.setTargetNormals(nrm)
            icp.setInputSource(src)
            icp.setMaximumIterations(25)
            icp.setMaxCorrespondenceDistance(md)
            icp.setTransformationEpsilon(1e-9)
            return icp
        dev, host = make(), make()
        host._ensure()
        conv = _lib.ConvergenceState()
        _lib.load().pclhip_convergence_init(C.byref(conv))
        for rep in range(2):   # twice: the criteria's MSE memory persists across align() calls
            dev.align()
            b = _host_loop_align(host, conv)
            assert dev.nr_iterations_ == b["it"] and dev.getConvergenceState() == states[b["state"]], (name, rep, b)
            assert dev.hasConverged() == b["conv"], (name, rep)
            assert np.abs(dev.getFinalTransformation() - b["T"]).max() < 2e-6, name
            assert np.abs(dev.getLastIncrementalTransformation() - b["last"]).max() < 2e-6, name


@pytest.fixture(scope="module")
def orc():
    from oracle import pcl_oracle
    return pcl_oracle


def test_rejector_chain_runs_inside_the_device_loop(gpu, orc):
    # MedianDistance / Trimmed / OneToOne / Distance between search and estimation, with every count, rank and threshold
    # kept in device memory (rejectors.hip): the device-driven alignment equals the oracle's loop over the same chain
    # (impl/icp.hpp:187-201), and runSteps replays it step for step
    import pcl_amd
    from oracle import rejectors as orej
    tgt, src, _ = pcl_amd.synth.icp_pair(40_000)

    def chain():
        a = pcl_amd.CorrespondenceRejectorMedianDistance(); a.setMedianFactor(1.5)
        b = pcl_amd.CorrespondenceRejectorTrimmed(); b.setOverlapRatio(0.8)
        c = pcl_amd.CorrespondenceRejectorOneToOne()
        d = pcl_amd.CorrespondenceRejectorDistance(); d.setMaximumDistance(0.05)
        return [a, b, c, d]

    icp = _make_icp(gpu, tgt, src, 0)
    for r in chain():
        icp.addCorrespondenceRejector(r)
    icp.align()
    assert icp.hasConverged()
    T_dev, it_dev = icp.getFinalTransformation().copy(), icp.nr_iterations_
    ref = orej.icp_with_filters(orc, tgt, src, 0,
                                rejectors=[lambda q, m, d: orej.reject_median_distance(q, m, d, 1.5),
                                           lambda q, m, d: orej.reject_trimmed(q, m, d, 0.8),
                                           orej.reject_one_to_one,
                                           lambda q, m, d: orej.reject_distance(q, m, d, 0.05)],
                                max_iterations=20, max_correspondence_distance=0.1, transformation_epsilon=1e-10)
    assert it_dev == ref["iterations"]
    assert np.abs(T_dev - ref["T"]).max() < 2e-5
    steps = icp.runSteps(it_dev)
    assert steps[-1]["alignment_ended"] and [s["iteration"] for s in steps] == list(range(1, it_dev + 1))
    assert np.abs(steps[-1]["final_transformation"] - T_dev).max() < 1e-7
    # reciprocal correspondences (impl/correspondence_estimation.hpp:220-311) on top of the chain: the source index is
    # built once and refitted to the moved cloud every iteration, all of it queued -- same answer as the oracle's loop
    icp.setUseReciprocalCorrespondences(True)
    icp.align()
    ref2 = orej.icp_with_filters(orc, tgt, src, 0, reciprocal=True,
                                 rejectors=[lambda q, m, d: orej.reject_median_distance(q, m, d, 1.5),
                                            lambda q, m, d: orej.reject_trimmed(q, m, d, 0.8),
                                            orej.reject_one_to_one,
                                            lambda q, m, d: orej.reject_distance(q, m, d, 0.05)],
                                 max_iterations=20, max_correspondence_distance=0.1, transformation_epsilon=1e-10)
    assert icp.nr_iterations_ == ref2["iterations"]
    assert np.abs(icp.getFinalTransformation() - ref2["T"]).max() < 2e-5
    steps = icp.runSteps(icp.nr_iterations_)
    assert steps[-1]["alignment_ended"]
    assert np.abs(steps[-1]["final_transformation"] - icp.getFinalTransformation()).max() < 1e-7


def test_run_steps_reports_no_correspondences(gpu):
    import pcl_amd
    tgt, src, _ = pcl_amd.synth.icp_pair(5_000)
    # nothing within reach: every step is an alignment that ends at once with NO_CORRESPONDENCES
    far = src.copy()
    far[:, 0] += 100.0
    icp2 = _make_icp(gpu, tgt, far, 0)
    steps = icp2.runSteps(3)
    assert [s["state"] for s in steps] == ["NO_CORRESPONDENCES"] * 3
    assert all(s["alignment_ended"] and not s["converged"] and s["iteration"] == 0 for s in steps)
    icp2.align()
    assert not icp2.hasConverged() and icp2.getConvergenceState() == "NO_CORRESPONDENCES" and icp2.nr_iterations_ == 0
    assert np.array_equal(icp2.getFinalTransformation(), np.eye(4, dtype=np.float32))


def test_native_communicator_single_rank(gpu):
    # the C-side RCCL path on a 1-rank group: ncclAllReduce from libpclhip.so on the context's stream;
    # a 1-rank sum is the identity, so results are bit-identical with and without the communicator
    import torch
    import pcl_amd
    from oracle import pcl_oracle as orc
    uid = pcl_amd.Communicator.unique_id()
    assert len(uid) == 128
    comm = pcl_amd.Communicator(gpu, 0, 1, uid)
    buf = torch.arange(32, dtype=torch.float64, device="cuda")
    comm.allreduce_sum_f64(buf.data_ptr(), 32)
    gpu.synchronize()
    assert torch.equal(buf.cpu(), torch.arange(32, dtype=torch.float64))
    tgt, src, _ = pcl_amd.synth.icp_pair(50_000)
    nrm = orc.KdTree(tgt).normals(tgt, 8, viewpoint=(0, 0, 10))[0]
    res = []
    for use in (False, True):
        icp = _make_icp(gpu, tgt, src, 1, nrm)
        if use:
            icp.setCommunicator(comm)
        icp.align()
        res.append((icp.getFinalTransformation().copy(), icp.nr_iterations_))
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]
    # MedianDistance / Trimmed under a communicator: the histograms of their selection are all-reduced (here over one rank:
    # the collective runs, the numbers are the unsharded chain's); OneToOne and reciprocal correspondences are refused
    # without a region (the source would be the sharded side) and run with one
    res = []
    for use in (False, True):
        icp = _make_icp(gpu, tgt, src, 1, nrm)
        a = pcl_amd.CorrespondenceRejectorMedianDistance()
        a.setMedianFactor(1.5)
        b = pcl_amd.CorrespondenceRejectorTrimmed()
        b.setOverlapRatio(0.8)
        d = pcl_amd.CorrespondenceRejectorDistance()
        d.setMaximumDistance(0.05)
        for r in (a, b, d):
            icp.addCorrespondenceRejector(r)
        if use:
            icp.setCommunicator(comm)
        icp.align()
        res.append((icp.getFinalTransformation().copy(), icp.nr_iterations_, len(icp.fetchCorrespondences()[0])))
        if use:
            assert np.array_equal(res[0][0], res[1][0]) and res[0][1:] == res[1][1:]
            icp.addCorrespondenceRejector(pcl_amd.CorrespondenceRejectorOneToOne())
            with pytest.raises(pcl_amd.PclHipError, match="OneToOne"):   # no region: the SOURCE would be the sharded side
                icp.align()
            # with the target sharded (one slab owning everything) the per-target minimum keys go through ncclAllReduce
            # (ncclUint64, ncclMin) -- over one rank the identity: the unsharded chain's result
            icp.setRegion([-np.inf] * 3 + [np.inf] * 3)
            icp.align()
            ref = _make_icp(gpu, tgt, src, 1, nrm)
            for r in (a, b, d, pcl_amd.CorrespondenceRejectorOneToOne()):
                ref.addCorrespondenceRejector(r)
            ref.align()
            assert icp.nr_iterations_ == ref.nr_iterations_
            assert np.array_equal(icp.getFinalTransformation(), ref.getFinalTransformation())
            assert len(icp.fetchCorrespondences()[0]) == len(ref.fetchCorrespondences()[0])
            icp2 = _make_icp(gpu, tgt, src, 1, nrm)
            icp2.setCommunicator(comm)
            icp2.setUseReciprocalCorrespondences(True)
            with pytest.raises(pcl_amd.PclHipError, match="whole source"):
                icp2.align()
            inf = np.inf
            icp2.setRegion([-inf] * 3 + [inf] * 3)       # the target sharded (here: one slab owning everything)
            icp2.align()
            one = _make_icp(gpu, tgt, src, 1, nrm)
            one.setUseReciprocalCorrespondences(True)
            one.align()
            assert icp2.nr_iterations_ == one.nr_iterations_
            assert np.array_equal(icp2.getFinalTransformation(), one.getFinalTransformation())
    assert 0 < res[0][2] < len(src)
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]


@pytest.mark.parametrize("deg", [5.0, 35.0])
def test_reciprocal_correspondences_after_a_rotation(gpu, deg):
    # the source index of the reciprocal test (impl/correspondence_estimation.hpp:247-270) is built in
    # the source's input frame and only REFITTED once the cloud has moved; after a rotation its nodes are not kd cells any
    # more, so a search of it must start at the root.  A guess of 5 / 35 degrees rotates the cloud before the first
    # search: every iteration's pair count and the final transform must be the oracle's.
    import pcl_amd
    from oracle import pcl_oracle as orc
    from oracle import rejectors as orej
    tgt, src0, _ = pcl_amd.synth.icp_pair(40_000)
    a = np.deg2rad(deg)
    G = np.eye(4, dtype=np.float32)
    G[:3, :3] = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
    Ginv = np.linalg.inv(G.astype(np.float64)).astype(np.float32)
    src = orc.transform_cloud(Ginv, src0, order=0)          # so that the guess brings the cloud back near the target
    icp = _make_icp(gpu, tgt, src, 0)
    icp.setUseReciprocalCorrespondences(True)
    icp.align(G)
    moved = orc.transform_cloud(G, src, order=0)
    ref = orej.icp_with_filters(orc, tgt, moved, 0, reciprocal=True, max_iterations=20,
                                max_correspondence_distance=0.1, transformation_epsilon=1e-10)
    assert icp.nr_iterations_ == ref["iterations"]
    T_ref = ref["T"].astype(np.float64) @ G.astype(np.float64)
    assert np.abs(icp.getFinalTransformation().astype(np.float64) - T_ref).max() < 2e-5
    steps = icp.runSteps(ref["iterations"], guess=G)
    assert [s["num_correspondences"] for s in steps] == [len(q) for q, _ in ref["per_iter"]][:len(steps)]


def test_device_driven_loop_correspondences_iteration_by_iteration(gpu, orc):
    # the full-size check of tests/test_gpu_fullsize.py (device-driven loop, every iteration's correspondences against
    # the oracle, the oracle's cloud moved by the device's own incremental transforms) at a size the CPU tier runs too
    import pcl_amd
    tgt, src, _ = pcl_amd.synth.icp_pair(150_000)
    otree = orc.KdTree(tgt)
    normals = otree.normals(tgt, 8, viewpoint=(0, 0, 10))[0]
    gpu.setOption("icp_lookahead", 0)
    try:
        cur = src.copy()
        full = None
        for K in range(1, 12):
            icp = _make_icp(gpu, tgt, src, 1, normals)   # a fresh object: the criteria keep their memory across align() calls
            icp.setMaximumIterations(K)
            icp.align()
            if icp.nr_iterations_ < K:
                full = icp.nr_iterations_
                break
            oq, om, od = otree.correspondences(cur, 0.1)
            q, m, d = icp.fetchCorrespondences()
            assert np.array_equal(q, oq) and np.array_equal(m, om) and np.array_equal(d.view(np.uint32), od.view(np.uint32)), K
            cur = orc.transform_cloud(icp.getLastIncrementalTransformation(), cur, order=1)
        assert full is not None and full >= 3
    finally:
        gpu.setOption("icp_lookahead", 1)
