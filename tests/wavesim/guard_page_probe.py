"""Strided inputs whose pointer sits INSIDE the caller's records (the normals of a PointNormal array: base + 16, stride 48)
are read to their last byte and no further: the array ends exactly at a page boundary and the next page is inaccessible, so
one byte too many is a segmentation fault.  (The three entry points below copied n * stride bytes from the interior pointer
until the sanitizer run of the C++ binding on the emulation found it.)  Run by tests/test_wavesim.py on the emulation; works
against the HIP library on a GPU box as well."""
import ctypes as C
import mmap
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pcl_amd  # noqa: E402
from pcl_amd import _lib  # noqa: E402

PAGE = mmap.PAGESIZE
libc = C.CDLL(None, use_errno=True)
libc.mmap.restype = C.c_void_p
libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]


def guarded(records):
    """a copy of `records` (float32, n x 12 = PointNormal) that ends exactly where an inaccessible page begins"""
    raw = np.ascontiguousarray(records, np.float32).tobytes()
    pages = (len(raw) + PAGE - 1) // PAGE
    base = libc.mmap(None, (pages + 1) * PAGE, mmap.PROT_READ | mmap.PROT_WRITE, mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, -1, 0)
    assert base not in (None, C.c_void_p(-1).value)
    assert libc.mprotect(C.c_void_p(base + pages * PAGE), PAGE, 0) == 0
    start = base + pages * PAGE - len(raw)
    C.memmove(C.c_void_p(start), raw, len(raw))
    return start


rng = np.random.default_rng(5)
n = 1000
rec = np.zeros((n, 12), np.float32)
rec[:, :3] = rng.uniform(-1, 1, (n, 3))
rec[:, 3] = 1
nrm = rng.normal(size=(n, 3))
rec[:, 4:7] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
tgt_p, src_p = guarded(rec), guarded(rec + np.float32(0.001))
lib = _lib.load()
ctx = pcl_amd.Context(0)
# TransformationEstimation on explicit pairs: points at +0, normals at +16 of the same records
T = (C.c_float * 16)()
for mode in (0, 1, 2):
    _lib.check(lib.pclhip_estimate_rigid_transformation(ctx.h, mode, C.c_void_p(src_p), 48, C.c_void_p(src_p + 16), 48,
                                                        C.c_void_p(tgt_p), 48, C.c_void_p(tgt_p + 16), 48, n, 1, T, None), ctx.h)
# the target's and the source's normals handed over as interior pointers
tree = pcl_amd.KdTree(ctx)
tree.setInputCloud(rec[:, :4].copy())
_lib.check(lib.pclhip_index_set_normals(tree.h, C.c_void_p(tgt_p + 16), 48), ctx.h)
icp = pcl_amd.IterativeClosestPointWithNormals(ctx)
icp.setSearchMethodTarget(tree, True)
icp.setInputSource(rec[:, :4].copy())
icp._ensure()
_lib.check(lib.pclhip_icp_set_source_normals(icp.h, C.c_void_p(src_p + 16), 48), ctx.h)
print("GUARD_PAGE ok")
