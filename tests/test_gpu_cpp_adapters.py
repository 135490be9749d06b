"""The C++ host mirror of the PCL plugin surface (include/pclhip/pcl_compat.hpp): compiled with plain
g++ against the C ABI (no HIP headers needed) and run against the reference's bunny goldens."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def link_args():
    """-L/-l/-rpath of the library the run is about: pcl_amd/libpclhip.so, or the build PCLHIP_LIB names (an A/B variant;
    tests/test_wavesim.py: the emulation of the CPU tier)."""
    lib = os.environ.get("PCLHIP_LIB") or os.path.join(ROOT, "pcl_amd", "libpclhip.so")
    d = os.path.dirname(os.path.abspath(lib))
    return ["-L" + d, "-l:" + os.path.basename(lib), "-Wl,-rpath," + d]


def build_cpp_test(tmp_path):
    exe = str(tmp_path / "test_pcl_compat")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_pcl_compat.cpp"), "-o", exe,
                           *link_args()])
    return exe


def test_cpp_adapters_compile_and_link(tmp_path):
    # CPU-only: the header-only adapters build with a plain C++17 compiler and link to the C ABI
    exe = build_cpp_test(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2  # usage error: no arguments -> nothing touched the GPU


@pytest.mark.gpu
def test_cpp_adapters_bunny_goldens(tmp_path, bunny, golden):
    exe = build_cpp_test(tmp_path)
    np.savetxt(tmp_path / "bun0.txt", bunny["bun0"][:, :3], fmt="%.9g")
    np.savetxt(tmp_path / "bun4.txt", bunny["bun4"][:, :3], fmt="%.9g")
    np.savetxt(tmp_path / "corr.txt", np.asarray(golden["correspondences_original"]), fmt="%d")
    r = subprocess.run([exe, str(tmp_path / "bun0.txt"), str(tmp_path / "bun4.txt"), str(tmp_path / "corr.txt")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ALL OK" in r.stdout


def build_c_example(tmp_path):
    exe = str(tmp_path / "icp_pcd")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "icp_pcd.c"), "-o", exe,
                           *link_args(), "-lm"])
    return exe


def test_c_example_compiles_as_plain_c(tmp_path):
    # CPU-only: include/pclhip.h is a C header; the example uses nothing but the C ABI
    exe = build_c_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
def test_c_example_registers_the_bunny_pair(tmp_path, bunny, golden):
    # PCD in -> index, normals, point-to-plane ICP, fitness -> PCD out, all through the C ABI from plain C
    import pcl_amd
    exe = build_c_example(tmp_path)
    src_path = os.path.join(ROOT, "tests", "golden", "pcd", "bun0.pcd")          # ascii, with normals (ignored)
    tgt_path = str(tmp_path / "bun4.pcd")
    tgt = np.ones((len(bunny["bun4"]), 4), np.float32)
    tgt[:, :3] = bunny["bun4"][:, :3]
    pcl_amd.savePCDFile(tgt_path, tgt, "binary_compressed")
    out_path = str(tmp_path / "aligned.pcd")
    r = subprocess.run([exe, src_path, tgt_path, out_path, "0.05", "50"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "converged 1" in r.stdout
    fitness = float(r.stdout.split("fitness ")[1].split(",")[0])
    assert fitness < 0.001                         # test/registration/test_registration.cpp:301-302
    aligned, dense = pcl_amd.loadPCDFile(out_path)
    assert dense and aligned.shape == (397, 4)
    # the written cloud is the source moved by the printed transformation
    rows = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("  ") and len(ln.split()) == 4]
    T = np.asarray(rows[-4:], np.float64)
    want = bunny["bun0"][:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    assert np.abs(aligned[:, :3] - want).max() < 1e-5


# ---- the real-PCL binding (include/pclhip/pcl_plugin.hpp) against the PCL mock -------------------------
def build_plugin_test(tmp_path):
    exe = str(tmp_path / "test_pcl_plugin")
    # -fopenmp: the mock's stock CorrespondenceEstimation runs PCL's own OpenMP loop over the search backend
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-fopenmp", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "tests", "cpp", "pcl_mock"),
                           os.path.join(ROOT, "tests", "cpp", "test_pcl_plugin.cpp"), "-o", exe,
                           *link_args()])
    return exe


def test_pcl_plugin_compiles_against_the_pcl_mock(tmp_path):
    # CPU-only: the subclasses of pcl::search::KdTree / CorrespondenceEstimationBase / IterativeClosestPoint[WithNormals]
    # shown in INTEGRATION.md are real code: they build against base classes with PCL's own signatures
    exe = build_plugin_test(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2  # usage error: nothing touched the GPU


@pytest.mark.gpu
def test_pcl_plugin_bunny_goldens_through_the_virtual_interfaces(tmp_path, bunny, golden):
    from oracle import pcl_oracle as orc
    exe = build_plugin_test(tmp_path)
    np.savetxt(tmp_path / "bun0.txt", bunny["bun0"][:, :3], fmt="%.9g")
    np.savetxt(tmp_path / "bun4.txt", bunny["bun4"][:, :3], fmt="%.9g")
    np.savetxt(tmp_path / "corr.txt", np.asarray(golden["correspondences_original"]), fmt="%d")
    tgt = np.ones((len(bunny["bun4"]), 4), np.float32)
    tgt[:, :3] = bunny["bun4"][:, :3]
    nrm, _ = orc.KdTree(tgt).normals(tgt, 10)
    np.savetxt(tmp_path / "bun4_normals.txt", nrm[:, :3], fmt="%.9g")
    r = subprocess.run([exe, str(tmp_path / "bun0.txt"), str(tmp_path / "bun4.txt"), str(tmp_path / "corr.txt"),
                        str(tmp_path / "bun4_normals.txt")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ALL OK" in r.stdout
