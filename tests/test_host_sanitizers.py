"""AddressSanitizer + UndefinedBehaviorSanitizer over the host-only code of libpclhip (CPU, no GPU, no HIP runtime).

pcd_io.cpp (the on-disk format either side of the path) and shard.cpp (slab partition, 8(e))
parse untrusted files / arbitrary clouds on the host.  tests/cpp/fuzz_host_io.cpp is compiled together with those two
sources by g++ with -fsanitize=address,undefined and run with a fixed seed: round trips through the three PCD encodings,
then mutants of those files and of the reference's own PCD files (hostile header values, flipped bytes, truncation,
inconsistent LZF size words) through every reader entry point; random / degenerate clouds through partition_slabs,
select_region and region_owner with the tiling property checked.  A reader may reject a mutant -- it may not crash, touch
memory out of bounds, overflow, leak, or spend seconds on a few hundred bytes (what a 4 GB size word used to cost).
"""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_pcd_io_and_slab_partition_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "fuzz_host_io")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), "-w", "-o", exe,
           os.path.join(ROOT, "tests", "cpp", "fuzz_host_io.cpp"), os.path.join(ROOT, "pcl_amd", "csrc", "pcd_io.cpp"),
           os.path.join(ROOT, "pcl_amd", "csrc", "shard.cpp"), "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and ("cannot find" in r.stderr or "libasan" in r.stderr or "libubsan" in r.stderr):
        pytest.skip("sanitizer runtimes not installed: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-3000:]
    goldens = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "pcd", "*.pcd")))
    assert goldens, "the reference's PCD fixtures are missing"
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, "1", "60", str(tmp_path)] + goldens, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "0 failed checks, 0 slow reads" in r.stdout, r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
