"""ctypes wrapper for the CPU oracle (oracle/pcl_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (pcl_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpcl_oracle.so")


def build(force=False):
    """Compile the C restatement with gcc (seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "pcl_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libpcl_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class Convergence(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("failure_after_max_iter", C.c_int),
                ("rotation_threshold", C.c_double), ("translation_threshold", C.c_double),
                ("mse_threshold_relative", C.c_double), ("mse_threshold_absolute", C.c_double),
                ("max_iterations_similar_transforms", C.c_int),
                ("iterations_similar_transforms", C.c_int),
                ("correspondences_prev_mse", C.c_double), ("correspondences_cur_mse", C.c_double),
                ("convergence_state", C.c_int)]


class IcpParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("max_correspondence_distance", C.c_double),
                ("transformation_epsilon", C.c_double),
                ("transformation_rotation_epsilon", C.c_double),
                ("euclidean_fitness_epsilon", C.c_double), ("min_number_correspondences", C.c_int),
                ("mode", C.c_int), ("acc_double", C.c_int), ("nthreads", C.c_int),
                ("use_reciprocal", C.c_int)]


class IcpResult(C.Structure):
    _fields_ = [("final_transformation", C.c_float * 16), ("nr_iterations", C.c_int),
                ("converged", C.c_int), ("convergence_state", C.c_int),
                ("last_num_correspondences", C.c_int64), ("last_mse", C.c_double),
                ("seconds_search", C.c_double), ("seconds_total", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        fp, ip, dp, vp = (C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                          C.c_void_p)
        L.orc_knn_bruteforce.restype = C.c_int
        L.orc_knn_bruteforce.argtypes = [fp, C.c_int64, C.c_int, fp, C.c_int64, C.c_int, C.c_int,
                                         ip, fp, C.c_int]
        L.orc_kdtree_build.restype = vp
        L.orc_kdtree_build.argtypes = [fp, C.c_int64, C.c_int]
        L.orc_kdtree_free.argtypes = [vp]
        L.orc_kdtree_size.restype = C.c_int64
        L.orc_kdtree_size.argtypes = [vp]
        L.orc_kdtree_knn.restype = C.c_int
        L.orc_kdtree_knn.argtypes = [vp, fp, C.c_int64, C.c_int, C.c_int, ip, fp, C.c_int]
        L.orc_correspondences.restype = C.c_int64
        L.orc_correspondences.argtypes = [vp, fp, C.c_int64, C.c_int, C.c_double, ip, ip, fp,
                                          C.c_int]
        L.orc_gicp_covariances.restype = C.c_int
        L.orc_gicp_covariances.argtypes = [vp, fp, C.c_int64, C.c_int, C.c_int, C.c_double, dp, C.c_int]
        L.orc_fitness_score.restype = C.c_double
        L.orc_fitness_score.argtypes = [vp, fp, C.c_int64, C.c_int, fp, C.c_double,
                                        C.POINTER(C.c_int64), C.c_int]
        L.orc_reciprocal_correspondences.restype = C.c_int64
        L.orc_reciprocal_correspondences.argtypes = [vp, vp, fp, C.c_int64, C.c_int, fp, C.c_int,
                                                     C.c_double, ip, ip, fp, C.c_int]
        L.orc_lls_point_to_plane.restype = C.c_int64
        L.orc_lls_point_to_plane.argtypes = [fp, C.c_int, fp, C.c_int, fp, C.c_int, ip, ip,
                                             C.c_int64, dp, fp]
        L.orc_lls_solve.argtypes = [dp, fp]
        L.orc_lls_symmetric.restype = C.c_int64
        L.orc_lls_symmetric.argtypes = [fp, C.c_int, fp, C.c_int, fp, C.c_int, fp, C.c_int, ip, ip,
                                        C.c_int64, C.c_int, C.c_int, dp, fp]
        L.orc_symmetric_solve.argtypes = [dp, fp]
        L.orc_umeyama.restype = C.c_int64
        L.orc_umeyama.argtypes = [fp, C.c_int, fp, C.c_int, ip, ip, C.c_int64, C.c_int, fp]
        L.orc_umeyama_from_sums.argtypes = [dp, C.c_double, fp]
        L.orc_transform_cloud.argtypes = [fp, C.c_int, fp, C.c_int, fp, C.c_int, fp, C.c_int, fp,
                                          C.c_int, C.c_int64]
        L.orc_mat4_mul.argtypes = [fp, fp, fp]
        L.orc_convergence_init.argtypes = [C.POINTER(Convergence)]
        L.orc_convergence_has_converged.restype = C.c_int
        L.orc_convergence_has_converged.argtypes = [C.POINTER(Convergence), C.c_int, fp, C.c_double]
        L.orc_icp_params_default.argtypes = [C.POINTER(IcpParams)]
        L.orc_icp_align.restype = C.c_int
        L.orc_icp_align.argtypes = [vp, fp, C.c_int, fp, C.c_int, fp, C.c_int64, C.c_int, fp,
                                    C.POINTER(IcpParams), C.POINTER(Convergence),
                                    C.POINTER(IcpResult), fp, ip]
        L.orc_mean_and_covariance.restype = C.c_uint
        L.orc_mean_and_covariance.argtypes = [fp, C.c_int, ip, C.c_int, fp, fp]
        L.orc_solve_plane_parameters.argtypes = [fp, fp, fp, fp, fp]
        L.orc_normals_knn.restype = C.c_int64
        L.orc_normals_knn.argtypes = [vp, fp, C.c_int64, C.c_int, C.c_int, fp, fp, ip, C.c_int]
        L.orc_normals_knn_indices.restype = C.c_int64
        L.orc_normals_knn_indices.argtypes = [vp, fp, C.c_int64, C.c_int, C.c_int, fp, ip, C.c_int64, fp, ip, C.c_int]
        L.orc_normals_knn_queries.restype = C.c_int64
        L.orc_normals_knn_queries.argtypes = [vp, fp, C.c_int, fp, C.c_int64, C.c_int, ip, C.c_int64, C.c_int, fp, fp, ip, C.c_int]
        L.orc_voxelgrid.restype = C.c_int64
        L.orc_voxelgrid.argtypes = [fp, C.c_int64, C.c_int, fp, C.c_uint, C.c_int, C.c_double,
                                    C.c_double, fp, ip]
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _cloud(a):
    """float32, C-contiguous, shape (n, s>=3); returns (array, n, stride_in_floats)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] >= 3
    return a, a.shape[0], a.shape[1]


def default_threads():
    return os.cpu_count() or 1


def knn_bruteforce(tgt, qry, k, nthreads=None):
    tgt, nt, ts = _cloud(tgt)
    qry, nq, qs = _cloud(qry)
    idx = np.empty((nq, k), np.int32)
    d2 = np.empty((nq, k), np.float32)
    lib().orc_knn_bruteforce(_f(tgt), nt, ts, _f(qry), nq, qs, k, _i(idx), _f(d2),
                             nthreads or default_threads())
    return idx, d2


class KdTree:
    """Exact kd-tree (the oracle's stand-in for pcl::KdTreeFLANN)."""

    def __init__(self, pts):
        self.pts, self.n, self.stride = _cloud(pts)
        self.h = lib().orc_kdtree_build(_f(self.pts), self.n, self.stride)

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:   # (at interpreter exit the module's globals may be gone already)
            try:
                lib().orc_kdtree_free(self.h)
            except Exception:
                pass
            self.h = None

    def size(self):
        return lib().orc_kdtree_size(self.h)

    def knn(self, qry, k, nthreads=None):
        qry, nq, qs = _cloud(qry)
        idx = np.empty((nq, k), np.int32)
        d2 = np.empty((nq, k), np.float32)
        lib().orc_kdtree_knn(self.h, _f(qry), nq, qs, k, _i(idx), _f(d2),
                             nthreads or default_threads())
        return idx, d2

    def correspondences(self, src, max_dist=np.sqrt(np.finfo(np.float64).max), nthreads=None):
        src, ns, ss = _cloud(src)
        q = np.empty(ns, np.int32)
        m = np.empty(ns, np.int32)
        d2 = np.empty(ns, np.float32)
        c = lib().orc_correspondences(self.h, _f(src), ns, ss, float(max_dist), _i(q), _i(m),
                                      _f(d2), nthreads or default_threads())
        return q[:c].copy(), m[:c].copy(), d2[:c].copy()

    def gicp_covariances(self, cloud, k=20, epsilon=0.001, nthreads=None):
        """GeneralizedIterativeClosestPoint::computeCovariances -> (n, 3, 3) float64"""
        cloud, n, cs = _cloud(cloud)
        out = np.empty((n, 9), np.float64)
        rc = lib().orc_gicp_covariances(self.h, _f(cloud), n, cs, int(k), float(epsilon), _d(out),
                                        nthreads or default_threads())
        if rc != 0:
            raise ValueError("Number or points in cloud is less than k_correspondences_")
        return out.reshape(n, 3, 3)

    def fitness_score(self, src, T, max_range=float(np.finfo(np.float64).max), nthreads=None):
        """Registration::getFitnessScore; returns (score, nr)."""
        src, ns, ss = _cloud(src)
        T = np.ascontiguousarray(T, np.float32).reshape(16)
        nr = C.c_int64(0)
        v = lib().orc_fitness_score(self.h, _f(src), ns, ss, _f(T), float(max_range), C.byref(nr),
                                    nthreads or default_threads())
        return float(v), int(nr.value)

    def reciprocal_correspondences(self, src_tree, src, tgt,
                                   max_dist=np.sqrt(np.finfo(np.float64).max), nthreads=None):
        src, ns, ss = _cloud(src)
        tgt, _, ts = _cloud(tgt)
        q = np.empty(ns, np.int32)
        m = np.empty(ns, np.int32)
        d2 = np.empty(ns, np.float32)
        c = lib().orc_reciprocal_correspondences(self.h, src_tree.h, _f(src), ns, ss, _f(tgt), ts,
                                                 float(max_dist), _i(q), _i(m), _f(d2),
                                                 nthreads or default_threads())
        return q[:c].copy(), m[:c].copy(), d2[:c].copy()

    def normals(self, cloud, k, viewpoint=(0.0, 0.0, 0.0), want_knn=False, nthreads=None, indices=None):
        """`cloud` must be the cloud the tree was built on (search surface == input); `indices` (Feature::
        setIndices) restricts the queries to cloud[indices], one output row per index."""
        cloud, n, cs = _cloud(cloud)
        assert n == self.n, "normals(): pass the tree's own cloud (use indices= for a subset)"
        ind = None if indices is None else np.ascontiguousarray(indices, np.int32)
        nq = n if ind is None else len(ind)
        out = np.empty((nq, 4), np.float32)
        knn = np.empty((nq, k), np.int32) if want_knn else None
        vp = np.asarray(viewpoint, np.float32)
        nan = lib().orc_normals_knn_indices(self.h, _f(cloud), n, cs, k, _f(vp),
                                            _i(ind) if ind is not None else None, nq, _f(out),
                                            _i(knn) if want_knn else None, nthreads or default_threads())
        return (out, knn, nan) if want_knn else (out, nan)

    def normals_at(self, surface, queries, k, viewpoint=(0.0, 0.0, 0.0), indices=None, nthreads=None):
        """Feature::setSearchSurface: the tree holds `surface`; one normal per query (or per queries[indices]),
        fitted to the k nearest SURFACE points, flipped towards the viewpoint as seen from the query."""
        surface, n, ss = _cloud(surface)
        assert n == self.n, "normals_at(): pass the cloud the tree was built on as the surface"
        queries, nqc, qs = _cloud(queries)
        ind = None if indices is None else np.ascontiguousarray(indices, np.int32)
        nq = nqc if ind is None else len(ind)
        out = np.empty((nq, 4), np.float32)
        vp = np.asarray(viewpoint, np.float32)
        nan = lib().orc_normals_knn_queries(self.h, _f(surface), ss, _f(queries), nqc, qs,
                                            _i(ind) if ind is not None else None, nq, int(k), _f(vp), _f(out), None,
                                            nthreads or default_threads())
        return out, nan


def lls_point_to_plane(src, tgt, nrm, q=None, m=None):
    src, ns, ss = _cloud(src)
    tgt, nt, ts = _cloud(tgt)
    nrm, _, ns_ = _cloud(nrm)
    n = len(q) if q is not None else ns
    qq = np.ascontiguousarray(q, np.int32) if q is not None else None
    mm = np.ascontiguousarray(m, np.int32) if m is not None else None
    sums = np.zeros(27, np.float64)
    T = np.zeros(16, np.float32)
    used = lib().orc_lls_point_to_plane(_f(src), ss, _f(tgt), ts, _f(nrm), ns_,
                                        _i(qq) if qq is not None else None,
                                        _i(mm) if mm is not None else None, n, _d(sums), _f(T))
    return T.reshape(4, 4), sums, used


def lls_symmetric(src, src_nrm, tgt, tgt_nrm, q=None, m=None, enforce_same_direction=True,
                  acc_double=True):
    """TransformationEstimationSymmetricPointToPlaneLLS; returns (T 4x4, sums27, used)."""
    src, ns, ss = _cloud(src)
    sn, _, sns = _cloud(src_nrm)
    tgt, nt, ts = _cloud(tgt)
    tn, _, tns = _cloud(tgt_nrm)
    n = len(q) if q is not None else ns
    qq = np.ascontiguousarray(q, np.int32) if q is not None else None
    mm = np.ascontiguousarray(m, np.int32) if m is not None else None
    sums = np.zeros(27, np.float64)
    T = np.zeros(16, np.float32)
    used = lib().orc_lls_symmetric(_f(src), ss, _f(sn), sns, _f(tgt), ts, _f(tn), tns,
                                   _i(qq) if qq is not None else None,
                                   _i(mm) if mm is not None else None, n,
                                   1 if enforce_same_direction else 0, 1 if acc_double else 0,
                                   _d(sums), _f(T))
    return T.reshape(4, 4), sums, used


def symmetric_solve(sums27):
    s = np.ascontiguousarray(sums27, np.float64)
    T = np.zeros(16, np.float32)
    lib().orc_symmetric_solve(_d(s), _f(T))
    return T.reshape(4, 4)


def lls_solve(sums27):
    s = np.ascontiguousarray(sums27, np.float64)
    T = np.zeros(16, np.float32)
    lib().orc_lls_solve(_d(s), _f(T))
    return T.reshape(4, 4)


def umeyama(src, tgt, q=None, m=None, acc_double=False):
    src, ns, ss = _cloud(src)
    tgt, nt, ts = _cloud(tgt)
    n = len(q) if q is not None else ns
    qq = np.ascontiguousarray(q, np.int32) if q is not None else None
    mm = np.ascontiguousarray(m, np.int32) if m is not None else None
    T = np.zeros(16, np.float32)
    lib().orc_umeyama(_f(src), ss, _f(tgt), ts, _i(qq) if qq is not None else None,
                      _i(mm) if mm is not None else None, n, int(acc_double), _f(T))
    return T.reshape(4, 4)


def umeyama_from_sums(sums15, count):
    s = np.ascontiguousarray(sums15, np.float64)
    T = np.zeros(16, np.float32)
    lib().orc_umeyama_from_sums(_d(s), float(count), _f(T))
    return T.reshape(4, 4)


def transform_cloud(T, pts, order=0, normals=None):
    T = np.ascontiguousarray(T, np.float32).reshape(16)
    pts, n, s = _cloud(pts)
    out = pts.copy()
    if normals is not None:
        nrm, _, ns_ = _cloud(normals)
        nout = nrm.copy()
        lib().orc_transform_cloud(_f(T), order, _f(pts), s, _f(out), s, _f(nrm), ns_, _f(nout),
                                  ns_, n)
        return out, nout
    lib().orc_transform_cloud(_f(T), order, _f(pts), s, _f(out), s, None, 0, None, 0, n)
    return out


def mat4_mul(A, B):
    A = np.ascontiguousarray(A, np.float32).reshape(16)
    B = np.ascontiguousarray(B, np.float32).reshape(16)
    Cm = np.zeros(16, np.float32)
    lib().orc_mat4_mul(_f(A), _f(B), _f(Cm))
    return Cm.reshape(4, 4)


def new_convergence():
    c = Convergence()
    lib().orc_convergence_init(C.byref(c))
    return c


def icp_align(tree, tgt, src, mode=0, tgt_normals=None, guess=None, conv=None, record=False,
              **kw):
    """IterativeClosestPoint(+WithNormals)::align on the oracle.  kw: IcpParams fields."""
    tgt, nt, ts = _cloud(tgt)
    src, ns, ss = _cloud(src)
    p = IcpParams()
    lib().orc_icp_params_default(C.byref(p))
    p.mode = mode
    p.nthreads = default_threads()
    for k_, v in kw.items():
        assert hasattr(p, k_), k_
        setattr(p, k_, v)
    if conv is None:
        conv = new_convergence()
    r = IcpResult()
    nrm = None
    tns = 0
    if tgt_normals is not None:
        nrm, _, tns = _cloud(tgt_normals)
    g = np.ascontiguousarray(guess, np.float32).reshape(16) if guess is not None else None
    per_T = np.zeros((p.max_iterations, 16), np.float32) if record else None
    per_m = np.full((p.max_iterations, ns), -2, np.int32) if record else None
    lib().orc_icp_align(tree.h, _f(tgt), ts, _f(nrm) if nrm is not None else None, tns, _f(src),
                        ns, ss, _f(g) if g is not None else None, C.byref(p), C.byref(conv),
                        C.byref(r), _f(per_T) if record else None, _i(per_m) if record else None)
    res = {
        "T": np.array(r.final_transformation, np.float32).reshape(4, 4),
        "iterations": r.nr_iterations, "converged": bool(r.converged),
        "state": r.convergence_state, "num_correspondences": r.last_num_correspondences,
        "mse": r.last_mse, "seconds_search": r.seconds_search, "seconds_total": r.seconds_total,
        "conv": conv,
    }
    if record:
        res["per_iter_T"] = per_T[:r.nr_iterations].reshape(-1, 4, 4)
        res["per_iter_match"] = per_m[:max(r.nr_iterations, 1)]
    return res


def mean_and_covariance(cloud, indices):
    cloud, n, cs = _cloud(cloud)
    idx = np.ascontiguousarray(indices, np.int32)
    cov = np.zeros(9, np.float32)
    cen = np.zeros(4, np.float32)
    cnt = lib().orc_mean_and_covariance(_f(cloud), cs, _i(idx), len(idx), _f(cov), _f(cen))
    return cov.reshape(3, 3), cen, cnt


def solve_plane_parameters(cov):
    cov = np.ascontiguousarray(cov, np.float32).reshape(9)
    out = [C.c_float() for _ in range(4)]
    lib().orc_solve_plane_parameters(_f(cov), *[C.byref(o) for o in out])
    return tuple(o.value for o in out)


def voxelgrid(cloud, leaf, min_points_per_voxel=0, limits=None, field=2, negative=False):
    """limits=(lo, hi): the pass-through filter on float `field` of the record (2 = z), `negative` = cut the inside"""
    cloud, n, cs = _cloud(cloud)
    leaf3 = np.asarray(leaf, np.float32)
    if leaf3.ndim == 0:
        leaf3 = np.full(3, leaf3, np.float32)
    out = np.empty((max(n, 1), 4), np.float32)
    ids = np.empty(max(n, 1), np.int32)
    has = limits is not None
    lo, hi = (limits if has else (0.0, 0.0))
    flags = (1 | (2 if negative else 0) | ((int(field) + 1) << 8)) if has else 0
    m = lib().orc_voxelgrid(_f(cloud), n, cs, _f(leaf3), int(min_points_per_voxel), flags,
                            float(lo), float(hi), _f(out), _i(ids))
    if m < 0:
        return None, None
    return out[:m].copy(), ids[:m].copy()
