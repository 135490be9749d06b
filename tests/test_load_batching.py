"""The streaming passes of the setup kernels keep their loads batched (CPU tier: reads the gfx950 code objects, no GPU).

A global load under a per-row guard (`if (i < n) x = a[i]`) in an unrolled loop is compiled as load, `s_waitcnt vmcnt(0)`,
use: the sixteen rows a thread handles of a 4096-element block become sixteen memory round trips one after the other
(round 6, `profiles/r06_experiments_ab.txt`, last section: `kd_load_box_kernel` 90 -> 59 us, `kp_count_kernel` 17 -> 12 us,
... at 10M points once the loads of a batch stood unconditionally in front of their first use).  Nothing in the results
depends on it, so only the instruction stream can tell when a later edit or a compiler change brings the chain back:
`scripts/load_wait_pattern.py` prints, per kernel, the order of loads / stores / waits and how often a load is followed by
`vmcnt(0)` at once ("Lw0xN").  This test holds the kernels that were fixed to a small N.

The passes replace nothing of the reference by themselves; they are steps of the index build
(kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp:99-136) and of VoxelGrid::applyFilter
(filters/include/pcl/filters/impl/voxel_grid.hpp:597-814).
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pcl_amd", "csrc")
TOOL = os.path.join(ROOT, "scripts", "load_wait_pattern.py")
LLVM = "/opt/rocm/lib/llvm/bin"

# kernel-name fragment -> most "load, then vmcnt(0) at once" pairs it may hold (what the tree has + headroom for scalar
# bookkeeping loads: a per-row chain shows as 8, 16 or 32 of them)
LIMITS = {
    "index_build.o": {
        "kp_count_kernel": 2,
        "kd_load_box_kernel": 4,
        "kp_hist_kernelILi1": 6,
        "kp_scatter_kernelILb1": 6,
        "kp_scatter_kernelILb0": 6,
    },
    "voxelgrid.o": {
        "vg_minmax_kernel": 2,
        "vg_key_kernel": 2,
        "vg_runcount_kernel": 2,
        "vg_runstart_kernel": 2,
        "rs_hist_kernel": 4,
        "rs_scatter_kernel": 4,
    },
    "rejectors.o": {
        "rs_hist_kernelILi0": 5,
        "rs_hist_kernelILi1": 5,
        "rs_hist_kernelILi2": 5,
    },
}


def _patterns(obj):
    r = subprocess.run([sys.executable, TOOL, obj], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = {}
    for line in r.stdout.splitlines():
        m = re.match(r"\S+\s+(\S+)\s+Lw0x(\d+)\s+(\S*)", line)
        if m:
            out[m.group(1)] = (int(m.group(2)), m.group(3))
    return out


@pytest.mark.parametrize("obj", sorted(LIMITS))
def test_streaming_passes_keep_their_loads_in_flight_together(obj):
    path = os.path.join(CSRC, obj)
    if not os.path.exists(path):
        pytest.skip("%s not built (run __graft_entry__.build())" % obj)
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("needs the ROCm llvm tools")
    pats = _patterns(path)
    assert pats, "no kernels found in %s" % obj
    for frag, limit in LIMITS[obj].items():
        hits = {k: v for k, v in pats.items() if frag in k}
        assert hits, "kernel %s not found in %s (renamed? update this test)" % (frag, obj)
        for name, (n, seq) in hits.items():
            assert n <= limit, ("%s: %d loads are waited for one by one (limit %d): a guarded load crept back into an "
                                "unrolled loop?  order of loads / waits: %s" % (name, n, limit, seq[:200]))
