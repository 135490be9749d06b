"""Kernel parity in the CPU tier: the HIP kernels themselves, run on a lane-accurate emulation of the wavefront.

tests/wavesim builds the sources of pcl_amd/csrc for the HOST (-DPCLHIP_WAVESIM: every lane a fiber, cross-lane
operations as rendezvous of the wavefront, global_load_lds / DPP / ballot semantics restated in tests/wavesim/wavesim.hpp,
the HIP runtime entry points over host memory) into tests/wavesim/libpclhip_wavesim.so.  That library is TEST
INFRASTRUCTURE: pcl_amd/_lib.py refuses it unless PCLHIP_ALLOW_WAVESIM=1, which only this module sets, in a subprocess;
bench.py and the product never see it, nothing is timed on it.  What it buys: the `-m gpu` parity tests -- the same test
functions, the same oracle -- run here, where no GPU exists, at the sizes the emulation finishes in seconds, so a kernel
change is checked against the oracle before it ever costs GPU minutes.  It does not replace the GPU run: code generation,
the hardware's DPP / LDS-DMA behaviour and anything about time are only checked there.

Selected: every algorithmic family of the path (k-NN register and heap kernels, normals incl. the recorded-leaf second
pass, seeded and stand-off ICP searches, device-driven loop with rejectors and reciprocal correspondences, radius
search, VoxelGrid, GICP covariances, slab regions) + the bounded fuzz slices and the degenerate inputs of the disc bounds.
Tests that need torch.cuda tensors or RCCL are left to the GPU tier.
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WS = os.path.join(ROOT, "tests", "wavesim")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

# no GPU memory, no communicator in the emulation
NOT_HERE = ("not torch_buffers and not pointnormal_and_downsample_all_data and not native_comm and not rccl "
            "and not communicator")


@pytest.fixture(scope="module")
def wavesim_lib():
    if not os.path.exists(CLANG) or shutil.which("make") is None:
        pytest.skip("needs the ROCm clang++ and make")
    r = subprocess.run(["make", "-C", WS, "-j", str(min(16, os.cpu_count() or 1))], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lib = os.path.join(WS, "libpclhip_wavesim.so")
    assert os.path.exists(lib)
    return lib


@pytest.fixture(scope="module")
def variant_lib(tmp_path_factory):
    """ONE more build of the emulation for the tests of build parameters: the served-group history capped at 3
    (-DPCLHIP_OWN_HIST_CAP=3) and the bounds checked where they cull (-DPCLHIP_VERIFY_BOUNDS, traverse.hpp)."""
    if not os.path.exists(CLANG) or shutil.which("make") is None:
        pytest.skip("needs the ROCm clang++ and make")
    build = tmp_path_factory.mktemp("ws_variant")
    mk = open(os.path.join(WS, "Makefile")).read()
    mk = mk.replace("SRC = ../../pcl_amd/csrc", "SRC = %s" % os.path.join(ROOT, "pcl_amd", "csrc"))
    mk = mk.replace("-I../../include", "-I" + os.path.join(ROOT, "include")).replace("../../include/pclhip.h", os.path.join(ROOT, "include", "pclhip.h"))
    (build / "Makefile").write_text(mk)
    for f in ("wavesim.hpp", "wavesim_rt.cpp", "pclhip_wave_reduce.hpp"):
        shutil.copy(os.path.join(WS, f), str(build / f))
    r = subprocess.run(["make", "-C", str(build), "-j", str(min(16, os.cpu_count() or 1)),
                        "EXTRA=-DPCLHIP_OWN_HIST_CAP=3 -DPCLHIP_VERIFY_BOUNDS"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return str(build / "libpclhip_wavesim.so")


def run_gpu_tests_on_the_emulation(lib, files, keyword, timeout=1500):
    env = dict(os.environ, PCLHIP_LIB=lib, PCLHIP_ALLOW_WAVESIM="1")
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "-k", keyword] + \
          [os.path.join(ROOT, "tests", f) for f in files]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=timeout)
    tail = (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and " failed" not in r.stdout, tail
    return r.stdout


def test_product_loader_refuses_the_emulation(wavesim_lib):
    # the emulation never stands in for the HIP library: without the test-only switch the loader raises
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['PCLHIP_LIB'] = %r; os.environ.pop('PCLHIP_ALLOW_WAVESIM', None)\n"
            "from pcl_amd import _lib\n"
            "try:\n    _lib.load()\nexcept _lib.PclHipUnavailable as e:\n    print('REFUSED', e)\n" % (ROOT, wavesim_lib))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert "REFUSED" in r.stdout and "no CPU fallback" in r.stdout, r.stdout + r.stderr


# left to the GPU tier (or to WAVESIM_FULL=1) only because of their run time on the emulation: 10-20 s each
SLOW = ("not sharded_target_on_device and not run_steps_is_align_repeated and not device_loop_matches_host_loop_twin "
        "and not sharded_bench_path and not radius_chunked_large and not 300001 and not 65553 and not surface_200k "
        "and not random_global_transforms and not fused_single_kernel and not cpp_adapters_bunny and not c_example_registers")


def test_gpu_parity_tests_run_on_the_emulation(wavesim_lib):
    """test_gpu_parity / test_gpu_loop / test_gpu_dist / test_gpu_fuzz / test_gpu_cpp_adapters (the C++ binding through PCL's
    virtuals on the mock, the compat mirror, the plain-C example), the same functions the GPU tier runs, against the
    oracle: k-NN (register and heap kernels, ties, NaNs, subsets, representations), normals (k and radius, search
    surfaces), seeded and stand-off ICP searches bit for bit, the device-driven loop with rejector chains and reciprocal
    correspondences, estimators, radius search, VoxelGrid, GICP covariances, slab regions, fuzz slices and the degenerate
    inputs of the disc bounds."""
    if (os.cpu_count() or 1) < 4 and os.environ.get("WAVESIM_FULL") != "1":
        pytest.skip("fewer than 4 cores: the emulation would take many minutes here (WAVESIM_FULL=1 runs it anyway)")
    keyword = NOT_HERE if os.environ.get("WAVESIM_FULL") == "1" else NOT_HERE + " and " + SLOW
    out = run_gpu_tests_on_the_emulation(
        wavesim_lib, ["test_gpu_parity.py", "test_gpu_loop.py", "test_gpu_dist.py", "test_gpu_fuzz.py",
                      "test_gpu_cpp_adapters.py", "test_gpu_lane.py"], keyword)
    last = [ln for ln in out.splitlines() if " passed" in ln][-1]
    print(last)
    assert int(last.split(" passed")[0].split()[-1]) >= 140, last


@pytest.mark.parametrize("mode,world", [("target", 2), ("source", 2), ("target", 3), ("target+rej", 2), ("source+rej", 2),
                                        ("target+recip", 2), ("source+rej+empty", 3), ("target+o2o", 2)])
def test_n_ranks_through_the_native_communicator(wavesim_lib, tmp_path, mode, world):
    """An N > 1 execution of the library's own multi-GPU code, which the one-GPU box of a round cannot give: `world`
    PROCESSES, each with its own (emulated) device, a native communicator created from one shared id (pclhip_comm_*; the
    collective underneath is the emulation's stand-in -- a sum in rank order through shared memory -- not RCCL), the
    device-driven loop all-reducing the record of every iteration between its reduction and its solve, regions masking
    the source per rank ("target") or the source cut into slabs ("source").  Checked: every rank ends with the SAME 4x4
    bit for bit (same all-reduced record, same solve), it is the single-process alignment's up to the summation order,
    the iteration counts agree, and what the ranks served adds up to the all-reduced count.  "+rej": a MedianDistance +
    Trimmed + Distance chain inside the loop -- the histograms of the two selections are all-reduced, so every rank cuts at
    the single-GPU run's thresholds; "+recip": reciprocal correspondences with the target sharded (the whole source on every
    rank, the served-group lists standing aside); "+empty": the last rank's share of the source is EMPTY -- it has nothing to
    filter but still issues the chain's histogram all-reduces, in step with its peers; "+o2o": the OneToOne
    rejector with the target sharded -- a target point that lies in two halos can be matched on two ranks, the smallest
    (distance, query) key over the ranks wins (a minimum all-reduce of the per-target keys), as on one GPU."""
    import numpy as np
    n = 60_000
    mode, _, extra = mode.partition("+")
    full_mode = mode + ("+" + extra if extra else "")
    work = str(tmp_path)
    import socket
    with socket.socket() as sk:   # a free port for the ranks' torch.distributed (gloo) rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PCLHIP_LIB=wavesim_lib, PCLHIP_ALLOW_WAVESIM="1", WAVESIM_THREADS="8",  # (target mode: the ranks walk their served groups)
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    worker = os.path.join(WS, "two_rank_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, full_mode, str(r), str(world), work, str(n)], env=env, cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    ranks = [np.load(os.path.join(work, "rank%d.npz" % r)) for r in range(world)]
    # single-process reference on the same emulation (no communicator, no region)
    extra_lines = ""
    extras = set(extra.split("+")) if extra else set()
    if "rej" in extras:
        extra_lines = ("a = pcl_amd.CorrespondenceRejectorMedianDistance(); a.setMedianFactor(1.5)\n"
                       "b = pcl_amd.CorrespondenceRejectorTrimmed(); b.setOverlapRatio(0.8)\n"
                       "d = pcl_amd.CorrespondenceRejectorDistance(); d.setMaximumDistance(0.05)\n"
                       "[icp.addCorrespondenceRejector(r) for r in (a, b, d)]\n")
    if "recip" in extras:
        extra_lines = "icp.setUseReciprocalCorrespondences(True)\n"
    if "o2o" in extras:
        extra_lines = "icp.addCorrespondenceRejector(pcl_amd.CorrespondenceRejectorOneToOne())\n"
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "import pcl_amd; from pcl_amd import synth\n"
            "tgt, src, _ = synth.icp_pair(%d); ctx = pcl_amd.Context(0)\n"
            "tree = pcl_amd.KdTree(ctx); tree.setInputCloud(tgt)\n"
            "ne = pcl_amd.NormalEstimation(ctx); ne.setInputCloud(tgt); ne.setSearchMethod(tree); ne.setKSearch(8)\n"
            "ne.setViewPoint(0, 0, 10); ne.compute(want_output=False)\n"
            "icp = pcl_amd.IterativeClosestPointWithNormals(ctx); icp.setSearchMethodTarget(tree, True); icp.setInputSource(src)\n"
            "icp.setMaximumIterations(20); icp.setMaxCorrespondenceDistance(0.1); icp.setTransformationEpsilon(1e-10)\n"
            "%s"
            "icp.align()\n"
            "kept = len(icp.fetchCorrespondences()[0])\n"
            "np.savez(%r, T=icp.getFinalTransformation(), iterations=icp.nr_iterations_, fitness=icp.getFitnessScore(0.01),\n"
            "         fitness_points=icp.fitness_points, kept_after_align=kept)\n" % (ROOT, n, extra_lines, os.path.join(work, "single.npz")))
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    one = np.load(os.path.join(work, "single.npz"))
    for r_ in ranks[1:]:
        assert np.array_equal(r_["T"], ranks[0]["T"]) and int(r_["iterations"]) == int(ranks[0]["iterations"])
        assert np.array_equal(r_["counts"], ranks[0]["counts"])            # all-reduced: the same on every rank
    assert int(ranks[0]["iterations"]) == int(one["iterations"]) and bool(ranks[0]["converged"])
    assert np.abs(ranks[0]["T"].astype(np.float64) - one["T"].astype(np.float64)).max() < 2e-6
    if "empty" in extras:
        assert int(ranks[-1]["served"]) == 0 and all(int(r_["served"]) > 0 for r_ in ranks[:-1])
    else:
        assert all(int(r_["served"]) > 0 for r_ in ranks)                   # every rank had work
    assert sum(int(r_["served"]) for r_ in ranks) in set(int(c) for c in ranks[0]["counts"])
    if mode == "target":
        # what the ranks kept in the alignment's last iteration adds up to what the single process kept: under the rejector
        # chains the cuts (medians, trim ranks, one-to-one winners) are the single-GPU run's, pair for pair
        assert sum(int(r_["kept_after_align"]) for r_ in ranks) == int(one["kept_after_align"])
    if not extra:
        assert int(ranks[0]["counts"][0]) == n
    else:
        assert 0 < int(ranks[0]["counts"][0]) < n                          # the chain / the reciprocal test dropped pairs
    if mode == "target":
        assert all(int(r_["index_points"]) < n for r_ in ranks)             # a slab + halo, not the cloud
        # getFitnessScore under sharding: the owned points' (sum, count) all-reduced -> the single-index score
        assert all(int(r_["fitness_points"]) == int(one["fitness_points"]) for r_ in ranks)
        assert all(abs(float(r_["fitness"]) - float(one["fitness"])) <= 1e-9 * float(one["fitness"]) for r_ in ranks)


def test_served_groups_when_an_alignment_outlasts_the_history(variant_lib, tmp_path):
    """The served-group lists of the sharded device loop remember the transforms of at most OWN_HIST_CAP launches (128);
    past that every group is served in every launch.  A build with a cap of 3 runs 12-iteration alignments through that
    path: served lists, matches, float distances and step records equal the full pass's (option "served_groups" 0)."""
    import numpy as np
    lib = variant_lib
    worker = os.path.join(ROOT, "tests", "owned_groups_worker.py")
    outs = []
    for owned in ("1", "0"):
        out = str(tmp_path / ("owned%s.npz" % owned))
        env = dict(os.environ, PCLHIP_LIB=lib, PCLHIP_ALLOW_WAVESIM="1")
        r = subprocess.run([sys.executable, worker, out, "60000", owned], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(np.load(out))
    a, b = outs
    assert int(a["strip_point_iterations"]) == 12          # four times the history
    for k in a.files:
        if k.endswith("_T"):
            assert np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max() < 2e-6, k
        elif k.endswith("_mse"):
            assert np.allclose(a[k], b[k], rtol=1e-9, atol=1e-18), k
        elif a[k].dtype == np.float32:
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
        else:
            assert np.array_equal(a[k], b[k]), k


def test_fuzz_slices_of_the_gpu_tier(wavesim_lib):
    """The bounded fuzz slices of the GPU tier (all ICP parity tests with WAVESIM_FULL=1) on the emulation."""
    lib = wavesim_lib
    files, keyword = ["test_gpu_fuzz.py"], NOT_HERE      # unseeded + seeded correspondences of random / degenerate clouds
    if os.environ.get("WAVESIM_FULL") == "1":             # every ICP-related parity test of the GPU tier (~45 s more)
        files = ["test_gpu_parity.py", "test_gpu_loop.py", "test_gpu_fuzz.py"]
        keyword = "(icp or rejector or reciprocal or fuzz or fitness) and " + NOT_HERE + " and " + SLOW
    out = run_gpu_tests_on_the_emulation(lib, files, keyword)
    print([ln for ln in out.splitlines() if " passed" in ln][-1])


@pytest.mark.skipif(os.environ.get("WAVESIM_SANITIZE") != "1",
                    reason="WAVESIM_SANITIZE=1: ~2 min of build + ~6 min of tests (profiles/r03_wavesim_sanitizers.txt is a run of it)")
def test_kernels_and_host_api_under_asan_ubsan(tmp_path):
    """The emulation built with -fsanitize=address,undefined (kernels, host API, emulation runtime) under the GPU tier's
    parity tests: a memory checker for the kernels' LDS / global accesses and the library's host code."""
    build = tmp_path / "ws_asan"
    build.mkdir()
    mk = open(os.path.join(WS, "Makefile")).read()
    mk = mk.replace("SRC = ../../pcl_amd/csrc", "SRC = %s" % os.path.join(ROOT, "pcl_amd", "csrc"))
    mk = mk.replace("-I../../include", "-I" + os.path.join(ROOT, "include")).replace("../../include/pclhip.h", os.path.join(ROOT, "include", "pclhip.h"))
    mk = mk.replace("$(CXX) -shared -fPIC -o $@", "$(CXX) -shared -shared-libasan -fsanitize=address,undefined -fPIC -o $@")
    mk = mk.replace("$(CXX) -std=c++17 -O2 -g -fPIC -fvisibility=hidden -D__HIP_PLATFORM_AMD__",
                    "$(CXX) -std=c++17 -O1 -g -fPIC -fvisibility=hidden -fsanitize=address,undefined -D__HIP_PLATFORM_AMD__")
    (build / "Makefile").write_text(mk)
    for f in ("wavesim.hpp", "wavesim_rt.cpp", "pclhip_wave_reduce.hpp"):
        shutil.copy(os.path.join(WS, f), str(build / f))
    r = subprocess.run(["make", "-C", str(build), "-j", str(min(16, os.cpu_count() or 1)),
                        "EXTRA=-fsanitize=address,undefined -fno-omit-frame-pointer"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    rt = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, PCLHIP_LIB=str(build / "libpclhip_wavesim.so"), PCLHIP_ALLOW_WAVESIM="1", LD_PRELOAD=rt,
               ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=0")
    cmd = [sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-k", NOT_HERE + " and not cpp_adapters and not c_example",
           os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_loop.py"),
           os.path.join(ROOT, "tests", "test_gpu_dist.py"), os.path.join(ROOT, "tests", "test_gpu_fuzz.py"),
           os.path.join(ROOT, "tests", "test_gpu_lane.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=3000)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    # the C++ host sides through the same build: the real-PCL binding on the mock and the compat mirror, with leak detection
    import json
    import numpy as np
    from oracle import pcl_oracle as orc
    z = np.load(os.path.join(ROOT, "tests", "golden", "bunny.npz"))
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
    np.savetxt(tmp_path / "bun0.txt", z["bun0"][:, :3], fmt="%.9g")
    np.savetxt(tmp_path / "bun4.txt", z["bun4"][:, :3], fmt="%.9g")
    np.savetxt(tmp_path / "corr.txt", np.asarray(g["correspondences_original"]), fmt="%d")
    tgt = np.ones((len(z["bun4"]), 4), np.float32)
    tgt[:, :3] = z["bun4"][:, :3]
    np.savetxt(tmp_path / "bun4_normals.txt", orc.KdTree(tgt).normals(tgt, 10)[0][:, :3], fmt="%.9g")
    common = [CLANG, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-shared-libasan", "-I" + os.path.join(ROOT, "include"),
              "-L" + str(build), "-l:libpclhip_wavesim.so", "-Wl,-rpath," + str(build), "-Wl,-rpath," + os.path.dirname(rt)]
    runs = [("test_pcl_plugin.cpp", ["-I" + os.path.join(ROOT, "tests", "cpp", "pcl_mock")], ["bun0.txt", "bun4.txt", "corr.txt", "bun4_normals.txt"]),
            ("test_pcl_compat.cpp", [], ["bun0.txt", "bun4.txt", "corr.txt"])]
    env2 = dict(os.environ, PCLHIP_ALLOW_WAVESIM="1", ASAN_OPTIONS="detect_leaks=1:detect_stack_use_after_return=0",
                UBSAN_OPTIONS="print_stacktrace=1")
    for src, extra, args in runs:
        exe = str(tmp_path / src.replace(".cpp", "_asan"))
        c = subprocess.run(common + extra + [os.path.join(ROOT, "tests", "cpp", src), "-o", exe], capture_output=True, text=True)
        assert c.returncode == 0, c.stderr[-3000:]
        r = subprocess.run([exe] + [str(tmp_path / a) for a in args], capture_output=True, text=True, env=env2, timeout=1800)
        assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]


@pytest.mark.skipif(os.environ.get("WAVESIM_FULL") != "1", reason="WAVESIM_FULL=1 (about a minute)")
def test_config2_at_its_full_size_on_the_emulation(wavesim_lib):
    """BASELINE.json's config 2 at its own size (2^20-point clouds, 20 point-to-point iterations): every iteration's
    correspondences bit for bit the oracle's and the SVD alignment within the 1e-5 contract -- the GPU tier's test, here."""
    out = run_gpu_tests_on_the_emulation(wavesim_lib, ["test_gpu_fullsize.py"], "config2")
    assert "1 passed" in out


def test_const_search_virtuals_from_eight_threads_under_tsan(wavesim_lib, tmp_path):
    """PCL calls the `const` search virtuals of ONE tree from OpenMP threads (impl/correspondence_estimation.hpp
    :163-175 with setNumberOfThreads, normal_3d_omp.hpp:76-81, impl/search.hpp:164-190).  tests/cpp/test_pcl_plugin.cpp section
    3b does that to a KdTreeHIP -- the mock's stock per-point CorrespondenceEstimation with 8 threads (397 golden pairs)
    and 8 plain threads mixing k-NN and radius searches -- here with the test and the binding (pcl_plugin.hpp) compiled
    -fsanitize=thread over the emulation.  (The OpenMP loop itself runs in the GPU tier and in the plain run of this
    file on the emulation; under TSan the test is built without -fopenmp, whose runtime is not instrumented, so the
    plain-thread half carries the concurrency.)  The library is not instrumented: its part of the contract -- the query
    entry points serialise on the context -- shows as equal results under every interleaving."""
    import json
    import numpy as np
    from oracle import pcl_oracle as orc
    z = np.load(os.path.join(ROOT, "tests", "golden", "bunny.npz"))
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
    np.savetxt(tmp_path / "bun0.txt", z["bun0"][:, :3], fmt="%.9g")
    np.savetxt(tmp_path / "bun4.txt", z["bun4"][:, :3], fmt="%.9g")
    np.savetxt(tmp_path / "corr.txt", np.asarray(g["correspondences_original"]), fmt="%d")
    tgt = np.ones((len(z["bun4"]), 4), np.float32)
    tgt[:, :3] = z["bun4"][:, :3]
    np.savetxt(tmp_path / "bun4_normals.txt", orc.KdTree(tgt).normals(tgt, 10)[0][:, :3], fmt="%.9g")
    exe = str(tmp_path / "test_pcl_plugin_tsan")
    d = os.path.dirname(os.path.abspath(wavesim_lib))
    c = subprocess.run([CLANG, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "tests", "cpp", "pcl_mock"), os.path.join(ROOT, "tests", "cpp", "test_pcl_plugin.cpp"),
                        "-o", exe, "-L" + d, "-l:" + os.path.basename(wavesim_lib), "-Wl,-rpath," + d], capture_output=True, text=True)
    if c.returncode != 0 and "tsan" in c.stderr.lower():
        pytest.skip("no ThreadSanitizer runtime for this clang")
    assert c.returncode == 0, c.stderr[-3000:]
    env = dict(os.environ, PCLHIP_ALLOW_WAVESIM="1", TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0")
    r = subprocess.run([exe] + [str(tmp_path / a) for a in ("bun0.txt", "bun4.txt", "corr.txt", "bun4_normals.txt")],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]


def test_strided_inputs_are_read_to_their_last_byte_only(wavesim_lib):
    """pclhip_estimate_rigid_transformation, pclhip_index_set_normals and pclhip_icp_set_source_normals take pointers INTO the
    caller's records (normals at record + 16, stride 48): arrays that end at an inaccessible page must not fault (they did:
    n * stride bytes were staged from the interior pointer, 16 past the array -- found by the sanitizer run of the C++
    binding on the emulation)."""
    env = dict(os.environ, PCLHIP_LIB=wavesim_lib, PCLHIP_ALLOW_WAVESIM="1")
    r = subprocess.run([sys.executable, os.path.join(WS, "guard_page_probe.py")], env=env, capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "GUARD_PAGE ok" in r.stdout, (r.returncode, r.stdout[-800:], r.stderr[-1500:])


def test_bounds_checked_where_they_cull_on_the_emulation(variant_lib):
    """tests/test_gpu_verify_bounds.py's worker on the emulation's -DPCLHIP_VERIFY_BOUNDS build: every leaf a disc bound or a
    reach filter drops has its true minimum distance evaluated by the lane it was dropped for; no claim may be broken, and
    at these sizes the results are compared with the oracle as well."""
    import json
    env = dict(os.environ, PCLHIP_LIB=variant_lib, PCLHIP_ALLOW_WAVESIM="1")
    fams = ["sheet", "cube", "collinear", "coincident", "far"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "verify_bounds_worker.py"), "30000"] + fams,
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    rows = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(rows) == 5 * len(fams)
    assert sum(x["checks"] for x in rows) > 1_000_000
    assert all(x["violations"] == 0 for x in rows), [x for x in rows if x["violations"]]
    assert all(x["matches_vs_oracle"] is True for x in rows)
