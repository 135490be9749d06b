"""ctypes binding of the C ABI in include/pclhip.h (the drop-in boundary).

The shared library is built in-tree by `__graft_entry__.build()` / `make -C pcl_amd/csrc`.  There is
deliberately NO CPU fallback: if the HIP library cannot be loaded, importing the product API fails
loudly with PclHipUnavailable.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCLHIP_LIB") or os.path.join(_HERE, "libpclhip.so")  # PCLHIP_LIB: A/B runs of two builds

NSUMS = 32
POINT_TO_POINT = 0
POINT_TO_PLANE = 1
SYMMETRIC = 2

CONVERGENCE_STATES = ("NOT_CONVERGED", "ITERATIONS", "TRANSFORM", "ABS_MSE", "REL_MSE",
                      "NO_CORRESPONDENCES", "FAILURE_AFTER_MAX_ITERATIONS")


class PclHipUnavailable(RuntimeError):
    pass


class PclHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("pclhip status %d: %s" % (status, message))
        self.status = status


class IcpParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("max_correspondence_distance", C.c_double),
                ("transformation_epsilon", C.c_double),
                ("transformation_rotation_epsilon", C.c_double),
                ("euclidean_fitness_epsilon", C.c_double), ("min_number_correspondences", C.c_int),
                ("mode", C.c_int), ("failure_after_max_iterations", C.c_int),
                ("max_iterations_similar_transforms", C.c_int),
                ("mse_threshold_absolute", C.c_double)]


class IcpResult(C.Structure):
    _fields_ = [("final_transformation", C.c_float * 16), ("last_transformation", C.c_float * 16),
                ("nr_iterations", C.c_int), ("converged", C.c_int), ("convergence_state", C.c_int),
                ("num_correspondences", C.c_uint64), ("mse", C.c_double), ("gpu_ms", C.c_double),
                ("gpu_ms_search_kernel", C.c_double)]


class IcpStep(C.Structure):
    _fields_ = [("iteration", C.c_int), ("convergence_state", C.c_int), ("converged", C.c_int),
                ("alignment_ended", C.c_int), ("num_correspondences", C.c_uint64), ("mse", C.c_double),
                ("search_ms", C.c_float), ("kernels_ms", C.c_float), ("step_ms", C.c_float),
                ("final_transformation", C.c_float * 16)]


class ConvergenceState(C.Structure):
    _fields_ = [("prev_mse", C.c_double), ("iterations_similar_transforms", C.c_int), ("convergence_state", C.c_int)]


class VoxelGridDims(C.Structure):
    _fields_ = [("min_b", C.c_int32 * 3), ("max_b", C.c_int32 * 3), ("div_b", C.c_int32 * 3), ("divb_mul", C.c_int32 * 3)]


COMM_ID_BYTES = 128


class Rejector(C.Structure):
    _fields_ = [("kind", C.c_int), ("param", C.c_double), ("min_correspondences", C.c_uint32),
                ("reserved", C.c_uint32)]


class PcdInfo(C.Structure):
    _fields_ = [("points", C.c_uint64), ("width", C.c_uint32), ("height", C.c_uint32), ("data_type", C.c_int),
                ("version", C.c_int), ("point_step", C.c_uint32), ("num_fields", C.c_uint32),
                ("has_xyz", C.c_int), ("has_normals", C.c_int), ("has_curvature", C.c_int),
                ("has_intensity", C.c_int), ("has_rgb", C.c_int), ("viewpoint", C.c_float * 7),
                ("data_offset", C.c_uint64)]


PCD_ASCII, PCD_BINARY, PCD_BINARY_COMPRESSED = 0, 1, 2

REJ_DISTANCE, REJ_MEDIAN_DISTANCE, REJ_ONE_TO_ONE, REJ_TRIMMED = 0, 1, 2, 3

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)

# every symbol include/pclhip.h declares: (restype, argtypes)
_vp, _sz, _u64 = C.c_void_p, C.c_size_t, C.c_uint64
SIGNATURES = {
    "pclhip_version": (C.c_char_p, []),
    "pclhip_last_error": (C.c_char_p, [_vp]),
    "pclhip_ctx_create": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "pclhip_ctx_create_on_stream": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "pclhip_ctx_stream": (_vp, [_vp]),
    "pclhip_ctx_destroy": (None, [_vp]),
    "pclhip_ctx_synchronize": (C.c_int, [_vp]),
    "pclhip_ctx_reserve": (C.c_int, [_vp, C.c_uint64]),
    "pclhip_ctx_set_option": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "pclhip_ctx_stats": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_uint64)]),
    "pclhip_index_build": (C.c_int, [_vp, _vp, _sz, _u64, _vp, _u64, C.POINTER(_vp)]),
    "pclhip_index_build_scaled": (C.c_int, [_vp, _vp, _sz, _u64, _vp, _u64, C.POINTER(C.c_float), C.POINTER(_vp)]),
    "pclhip_index_build_ex": (C.c_int, [_vp, _vp, _sz, _u64, _vp, _u64, C.POINTER(C.c_float), _sz, C.POINTER(_vp)]),
    "pclhip_icp_transform_source": (C.c_int, [_vp, C.POINTER(C.c_float), C.c_int, _vp, _vp, _sz, _u64, _sz]),
    "pclhip_index_destroy": (None, [_vp]),
    "pclhip_index_size": (_u64, [_vp]),
    "pclhip_index_build_ms": (C.c_double, [_vp]),
    "pclhip_index_order": (C.c_int, [_vp, _vp]),
    "pclhip_index_cells": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "pclhip_knn": (C.c_int, [_vp, _vp, _sz, _u64, C.c_int, _vp, _vp]),
    "pclhip_radius_search": (C.c_int, [_vp, _vp, _sz, _u64, C.c_double, C.c_uint32, C.POINTER(_u64), _vp, _vp, _u64,
                                       C.POINTER(_u64)]),
    "pclhip_normals": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_float), _vp, _sz, C.POINTER(_u64)]),
    "pclhip_normals_records": (C.c_int, [_vp, C.c_int, C.c_double, C.POINTER(C.c_float), _vp, _sz, _sz, _sz, C.POINTER(_u64)]),
    "pclhip_normals_radius": (C.c_int, [_vp, C.c_double, C.POINTER(C.c_float), _vp, _sz, C.POINTER(_u64)]),
    "pclhip_normals_at": (C.c_int, [_vp, _vp, _sz, _u64, _vp, _u64, C.c_int, C.c_double, C.POINTER(C.c_float), _vp, _sz,
                                    C.POINTER(_u64)]),
    "pclhip_gicp_covariances": (C.c_int, [_vp, C.c_int, C.c_double, _vp]),
    "pclhip_index_set_normals": (C.c_int, [_vp, _vp, _sz]),
    "pclhip_icp_params_default": (None, [C.POINTER(IcpParams)]),
    "pclhip_icp_create": (C.c_int, [_vp, C.POINTER(_vp)]),
    "pclhip_icp_destroy": (None, [_vp]),
    "pclhip_icp_set_source": (C.c_int, [_vp, _vp, _sz, _u64]),
    "pclhip_icp_set_source_indexed": (C.c_int, [_vp, _vp, _sz, _u64, _vp, _u64]),
    "pclhip_icp_set_source_normals": (C.c_int, [_vp, _vp, _sz]),
    "pclhip_icp_set_enforce_same_direction_normals": (C.c_int, [_vp, C.c_int]),
    "pclhip_icp_set_allreduce": (C.c_int, [_vp, ALLREDUCE_FN, _vp]),
    "pclhip_icp_reset": (C.c_int, [_vp]),
    "pclhip_icp_set_rejectors": (C.c_int, [_vp, C.POINTER(Rejector), C.c_int]),
    "pclhip_icp_last_median_distance": (C.c_double, [_vp]),
    "pclhip_icp_set_reciprocal": (C.c_int, [_vp, C.c_int]),
    "pclhip_icp_iterate": (C.c_int, [_vp, C.POINTER(C.c_float), C.c_double, C.c_int,
                                     C.POINTER(C.c_double)]),
    "pclhip_icp_last_kernel_ms": (C.c_double, [_vp]),
    "pclhip_icp_last_search_ms": (C.c_double, [_vp]),
    "pclhip_icp_source_order_ms": (C.c_double, [_vp]),
    "pclhip_index_last_kernel_ms": (C.c_double, [_vp]),
    "pclhip_solve_transformation": (C.c_int, [C.POINTER(C.c_double), C.c_int,
                                              C.POINTER(C.c_float)]),
    "pclhip_icp_align": (C.c_int, [_vp, C.POINTER(IcpParams), C.POINTER(C.c_float),
                                   C.POINTER(IcpResult)]),
    "pclhip_convergence_init": (None, [C.POINTER(ConvergenceState)]),
    "pclhip_convergence_has_converged": (C.c_int, [C.POINTER(IcpParams), C.POINTER(ConvergenceState), C.c_int,
                                                   C.POINTER(C.c_float), C.c_double]),
    "pclhip_icp_run_steps": (C.c_int, [_vp, C.POINTER(IcpParams), C.POINTER(C.c_float), C.c_int, C.POINTER(IcpStep)]),
    "pclhip_comm_get_unique_id": (C.c_int, [C.POINTER(C.c_ubyte)]),
    "pclhip_comm_create": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.POINTER(_vp)]),
    "pclhip_comm_destroy": (None, [_vp]),
    "pclhip_comm_rank": (C.c_int, [_vp]),
    "pclhip_comm_size": (C.c_int, [_vp]),
    "pclhip_comm_allreduce_sum_f64": (C.c_int, [_vp, _vp, C.c_int]),
    "pclhip_icp_set_comm": (C.c_int, [_vp, _vp]),
    "pclhip_partition_slabs": (C.c_int, [_vp, _sz, _u64, C.c_int, C.POINTER(C.c_float)]),
    "pclhip_select_region": (C.c_int, [_vp, _sz, _u64, C.POINTER(C.c_float), C.c_double, _vp, _u64, C.POINTER(_u64)]),
    "pclhip_region_owner": (C.c_int, [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float)]),
    "pclhip_icp_set_region": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "pclhip_index_kth_distance_max": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)]),
    "pclhip_icp_fitness_score": (C.c_int, [_vp, C.POINTER(C.c_float), C.c_double, C.POINTER(C.c_double),
                                           C.POINTER(_u64)]),
    "pclhip_icp_fetch_correspondence_records": (C.c_int, [_vp, _vp, _u64, C.POINTER(_u64)]),
    "pclhip_icp_fetch_correspondences": (C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(_u64)]),
    "pclhip_estimate_rigid_transformation": (C.c_int, [_vp, C.c_int, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _u64,
                                                       C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)]),
    "pclhip_estimate_rigid_transformation_weighted": (C.c_int, [_vp, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _u64,
                                                                C.POINTER(C.c_float), C.POINTER(C.c_double)]),
    "pclhip_transform_cloud": (C.c_int, [_vp, C.POINTER(C.c_float), C.c_int, _vp, _vp, _sz, _u64,
                                         _sz]),
    "pclhip_pcd_read_header": (C.c_int, [C.c_char_p, C.POINTER(PcdInfo)]),
    "pclhip_pcd_read": (C.c_int, [C.c_char_p, _vp, _sz, _sz, _u64, C.POINTER(_u64), C.POINTER(C.c_int)]),
    "pclhip_pcd_write": (C.c_int, [C.c_char_p, _vp, _sz, _sz, _u64, C.c_int, C.c_int]),
    "pclhip_pcd_write_organized": (C.c_int, [C.c_char_p, _vp, _sz, _sz, C.c_uint32, C.c_uint32, C.POINTER(C.c_float),
                                             C.c_int, C.c_int]),
    "pclhip_pcd_read_field": (C.c_int, [C.c_char_p, C.c_char_p, C.c_uint32, _vp, _u64, C.POINTER(_u64)]),
    "pclhip_voxelgrid_ex": (C.c_int, [_vp, _vp, _sz, _u64, C.POINTER(C.c_float), C.c_uint32, C.c_int, C.c_double,
                                      C.c_double, C.c_int, _sz, _vp, _sz, C.POINTER(_u64)]),
    "pclhip_voxelgrid_ex2": (C.c_int, [_vp, _vp, _sz, _u64, C.POINTER(C.c_float), C.c_uint32, C.c_int, C.c_double,
                                       C.c_double, C.c_int, _sz, _vp, _sz, C.POINTER(_u64), _vp, _u64,
                                       C.POINTER(VoxelGridDims)]),
    "pclhip_voxelgrid_grid": (C.c_int, [_vp, _vp, _sz, _u64, C.POINTER(C.c_float), C.c_int, C.c_double, C.c_double,
                                        C.POINTER(VoxelGridDims)]),
    "pclhip_voxelgrid": (C.c_int, [_vp, _vp, _sz, _u64, C.POINTER(C.c_float), C.c_uint32, C.c_int,
                                   C.c_double, C.c_double, _vp, C.POINTER(_u64)]),
}

_lib = None


def load():
    """dlopen libpclhip.so and bind every declared symbol (no compute, works without a GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PclHipUnavailable(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C pcl_amd/csrc`).  There is no CPU fallback." % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # missing libamdhip64 etc.
        raise PclHipUnavailable("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    # tests/wavesim builds the same sources for the host (a lane-accurate emulation of the wavefront, used by the CPU test
    # tier to check kernels against the oracle).  It is test infrastructure: the product never runs on it.
    if b"wavesim" in (lib.pclhip_version() or b"") and os.environ.get("PCLHIP_ALLOW_WAVESIM") != "1":
        raise PclHipUnavailable("%s is the CPU emulation of the test tier (tests/wavesim), not the HIP library; only "
                                "tests/test_wavesim.py loads it (PCLHIP_ALLOW_WAVESIM=1).  There is no CPU fallback." % LIB_PATH)
    _lib = lib
    return lib


def check(status, ctx_handle=None):
    if status != 0:
        msg = load().pclhip_last_error(ctx_handle)
        raise PclHipError(status, msg.decode() if msg else "")
