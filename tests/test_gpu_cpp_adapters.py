"""The C++ host mirror of the PCL plugin surface (include/pclhip/pcl_compat.hpp): compiled with plain
g++ against the C ABI (no HIP headers needed) and run against the reference's bunny goldens."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_cpp_test(tmp_path):
    exe = str(tmp_path / "test_pcl_compat")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_pcl_compat.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "pcl_amd"), "-lpclhip",
                           "-Wl,-rpath," + os.path.join(ROOT, "pcl_amd")])
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
