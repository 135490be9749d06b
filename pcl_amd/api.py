"""Python mirror of the PCL plugin surface for the ICP hot path, over the C ABI (include/pclhip.h).

Class and method names follow the reference so parity tests read like PCL's own tests:
  pcl::search::KdTree<PointT>                         search/include/pcl/search/kdtree.h:61-168
  pcl::registration::CorrespondenceEstimation          registration/include/pcl/registration/correspondence_estimation.h
  pcl::IterativeClosestPoint / ...WithNormals          registration/include/pcl/registration/icp.h:98-347,360-440
  pcl::NormalEstimation                                features/include/pcl/features/normal_3d.h:243-420
  pcl::VoxelGrid                                       filters/include/pcl/filters/voxel_grid.h:221-533
Clouds are (n, c>=3) float32 arrays: numpy (host) or torch CUDA tensors (device-resident; only the
pointer crosses the boundary).  All compute happens in libpclhip.so; there is no CPU fallback.
"""
import ctypes as C
import os
import math

import numpy as np

from . import _lib
from ._lib import POINT_TO_PLANE, POINT_TO_POINT, IcpParams, IcpResult, check

_SQRT_DBL_MAX = math.sqrt(np.finfo(np.float64).max)


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _cloud(a):
    """-> (pointer, stride_bytes, n, keepalive)."""
    if _is_torch(a):
        assert a.dtype.is_floating_point and a.element_size() == 4 and a.dim() == 2 and a.shape[1] >= 3
        a = a.contiguous()
        return C.c_void_p(a.data_ptr()), a.shape[1] * 4, a.shape[0], a
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] >= 3
    return C.c_void_p(a.ctypes.data), a.shape[1] * 4, a.shape[0], a


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Context:
    """One device + one HIP stream (pclhip_ctx)."""

    def __init__(self, device=0, stream=None):
        """stream=None: the context creates its own non-blocking stream.  Any integer -- including 0,
        the legacy default stream torch reports as current_stream().cuda_stream -- is ADOPTED as given,
        so work the caller orders on that stream (collectives, copies) stays ordered with the kernels."""
        self.lib = _lib.load()
        h = C.c_void_p()
        if stream is None:
            check(self.lib.pclhip_ctx_create(int(device), None, C.byref(h)))
        else:
            check(self.lib.pclhip_ctx_create_on_stream(int(device), C.c_void_p(int(stream)), C.byref(h)))
        self.h = h
        self.device = device
        self.stream = int(self.lib.pclhip_ctx_stream(h) or 0)
        self._children = []  # weakrefs to objects holding handles that point into this context

    def _adopt(self, obj):
        import weakref
        self._children.append(weakref.ref(obj))

    def synchronize(self):
        check(self.lib.pclhip_ctx_synchronize(self.h), self.h)

    def reserve(self, nbytes):
        """Reserve the context's device arena up front (pclhip_ctx_reserve); automatic for the first large cloud otherwise."""
        check(self.lib.pclhip_ctx_reserve(self.h, int(nbytes)), self.h)

    def setOption(self, name, value):
        """pclhip_ctx_set_option: "served_groups", "icp_lookahead", "cache_mb", "arena_mb", "lane_search", "lane_max_up",
        "lane_far" (none changes a result)."""
        check(self.lib.pclhip_ctx_set_option(self.h, name.encode(), float(value)), self.h)

    def stats(self, enable=True):
        """Read (then re-arm or disable) the traversal work counters."""
        out = (C.c_uint64 * 8)()
        check(self.lib.pclhip_ctx_stats(self.h, int(enable), out), self.h)
        names = ("nodes", "leaves_group", "leaves_allpairs", "pushes", "groups", "so_done", "so_list", "so_union")
        return {n: int(out[i]) for i, n in enumerate(names)}

    def counters(self, enable=True):
        """The same eight counters as a plain list (the per-lane search kernels of lane.hip count in them too: 0 queries,
        1 done with their own leaf, 2 done in the first pass, 3 greedy descents, 4 finished by the second pass)."""
        out = (C.c_uint64 * 8)()
        check(self.lib.pclhip_ctx_stats(self.h, int(enable), out), self.h)
        return [int(v) for v in out]

    def close(self):
        if getattr(self, "h", None):
            # handles hold a raw pointer to the context: release them first, whatever order the
            # interpreter finalises objects in
            for ref in self._children:
                obj = ref()
                if obj is not None:
                    obj._release()
            self._children = []
            self.lib.pclhip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Communicator:
    """pclhip_comm: an RCCL group (one rank per GPU/process) for the per-iteration all-reduce of the record.
    `unique_id()` on rank 0, distribute the 128 bytes (e.g. torch.distributed.broadcast_object_list over gloo,
    MPI, a file), then Communicator(ctx, rank, nranks, id) on every rank."""

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * _lib.COMM_ID_BYTES)()
        check(_lib.load().pclhip_comm_get_unique_id(buf))
        return bytes(buf)

    def __init__(self, ctx, rank, nranks, uid):
        self.ctx = ctx
        self.lib = ctx.lib
        assert len(uid) == _lib.COMM_ID_BYTES
        buf = (C.c_ubyte * _lib.COMM_ID_BYTES).from_buffer_copy(uid)
        h = C.c_void_p()
        check(self.lib.pclhip_comm_create(ctx.h, int(rank), int(nranks), buf, C.byref(h)), ctx.h)
        self.h = h
        self.rank, self.size = int(rank), int(nranks)
        ctx._adopt(self)

    def allreduce_sum_f64(self, device_ptr, count):
        check(self.lib.pclhip_comm_allreduce_sum_f64(self.h, C.c_void_p(int(device_ptr)), int(count)), self.ctx.h)

    def _release(self):
        if getattr(self, "h", None):
            if self.ctx.h is not None:
                self.lib.pclhip_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class KdTree:
    """pcl::search::KdTree<PointT> backed by the GPU kd-ordered wide BVH (pclhip_index)."""

    def __init__(self, ctx=None, sorted_results=True):
        self.ctx = ctx or default_context()
        self.lib = self.ctx.lib
        self.h = None
        self._cloud_id = None
        self._cloud = None
        self._indices = None
        self.n_cloud = 0
        self._sorted = bool(sorted_results)
        self._epsilon = 0.0
        self.ctx._adopt(self)

    def _release(self):
        self._free()

    # search/include/pcl/search/search.h:96-142, search/kdtree.h:110-143
    def getName(self):
        return "KdTree"

    def getInputCloud(self):
        return self._cloud

    def getIndices(self):
        return self._indices

    def setSortedResults(self, sorted_results):
        """search.h:104-110: results always come back ascending by distance here, which satisfies either setting."""
        self._sorted = bool(sorted_results)

    def getSortedResults(self):
        return self._sorted

    def setEpsilon(self, eps):
        """search/kdtree.h:130-143: FLANN's (1 + eps) approximate search; this index always answers exactly."""
        self._epsilon = float(eps)

    def getEpsilon(self):
        return self._epsilon

    def setInputCloud(self, cloud, indices=None):
        # search/include/pcl/search/impl/kdtree.hpp:87-97 -> kdtree_flann.hpp:99-136: ALWAYS rebuilds, like
        # the reference (the array may have been modified in place since the last call).  Callers that know
        # the cloud is unchanged keep the tree and pass it with setSearchMethodTarget(tree, True).
        self._cloud, self._indices = cloud, indices
        key = (id(cloud), None if indices is None else id(indices))
        self._free()
        ptr, stride, n, keep = _cloud(cloud)
        h = C.c_void_p()
        ind = None if indices is None else np.ascontiguousarray(indices, np.int32)
        scale = getattr(self, "_scale", None)
        check(self.lib.pclhip_index_build_scaled(self.ctx.h, ptr, stride, n,
                                                 None if ind is None else C.c_void_p(ind.ctypes.data),
                                                 0 if ind is None else len(ind),
                                                 None if scale is None else _fp(scale), C.byref(h)), self.ctx.h)
        self.h = h
        self._cloud_id = key
        self._keep = (cloud, indices)
        self.n_cloud = n
        return True

    def setPointRepresentation(self, rescale_values=None, dimensions=3):
        """pcl::search::KdTree::setPointRepresentation (search/include/pcl/search/kdtree.h:110) for the
        representations the 3-D index can honour: the first `dimensions` of (x, y, z), each times its rescale
        value (CustomPointRepresentation(dimensions) + setRescaleValues, common/include/pcl/point_representation.h
        :150-190,546-579).  None / 3 = the default representation.  Takes effect at the next setInputCloud."""
        if rescale_values is None and dimensions == 3:
            self._scale = None
            return
        assert 1 <= dimensions <= 3, "the index is three-dimensional: only prefixes of (x, y, z)"
        sc = np.zeros(3, np.float32)
        rv = np.ones(dimensions, np.float32) if rescale_values is None else np.asarray(rescale_values, np.float32)
        assert rv.shape == (dimensions,)
        sc[:dimensions] = rv
        self._scale = sc

    def kthDistanceMax(self, k, box=None):
        """pclhip_index_kth_distance_max: the largest k-th-neighbour distance (not squared) over the indexed
        points inside `box` (lo.xyz, hi.xyz; None: all)."""
        out = C.c_double(0.0)
        b = None if box is None else np.ascontiguousarray(box, np.float32).reshape(6)
        check(self.lib.pclhip_index_kth_distance_max(self.h, int(k), _fp(b) if b is not None else None, C.byref(out)),
              self.ctx.h)
        return math.sqrt(out.value)

    def size(self):
        return int(self.lib.pclhip_index_size(self.h))

    def build_ms(self):
        return float(self.lib.pclhip_index_build_ms(self.h))

    def order(self):
        """pclhip_index_order: original index of the point at every position of the index's kd order (int32 array)."""
        out = np.empty(self.size(), np.int32)
        if len(out):
            check(self.lib.pclhip_index_order(self.h, C.c_void_p(out.ctypes.data)), self.ctx.h)
        return out

    def cells(self, level):
        """pclhip_index_cells: (boxes [n,6], cells [n,6], top level) of one quad level of the per-lane search structure."""
        cnt, top = C.c_uint64(0), C.c_int(0)
        check(self.lib.pclhip_index_cells(self.h, int(level), None, None, 0, C.byref(cnt), C.byref(top)), self.ctx.h)
        boxes = np.empty((cnt.value, 6), np.float32)
        cells = np.empty((cnt.value, 6), np.float32)
        check(self.lib.pclhip_index_cells(self.h, int(level), C.c_void_p(boxes.ctypes.data), C.c_void_p(cells.ctypes.data),
                                          cnt.value, C.byref(cnt), C.byref(top)), self.ctx.h)
        return boxes, cells, top.value

    def lastKernelMs(self):
        return float(self.lib.pclhip_index_last_kernel_ms(self.h))

    def nearestKSearch(self, queries, k, out=None):
        """Batch overload (search.h:216-219).  Returns (indices int32 [nq,k], sqr_distances [nq,k])."""
        ptr, stride, nq, keep = _cloud(queries)
        if _is_torch(queries):
            import torch
            idx = torch.empty((nq, k), dtype=torch.int32, device=queries.device)
            d2 = torch.empty((nq, k), dtype=torch.float32, device=queries.device)
            check(self.lib.pclhip_knn(self.h, ptr, stride, nq, int(k), C.c_void_p(idx.data_ptr()),
                                      C.c_void_p(d2.data_ptr())), self.ctx.h)
            return idx, d2
        idx = np.empty((nq, k), np.int32)
        d2 = np.empty((nq, k), np.float32)
        check(self.lib.pclhip_knn(self.h, ptr, stride, nq, int(k), C.c_void_p(idx.ctypes.data),
                                  C.c_void_p(d2.ctypes.data)), self.ctx.h)
        return idx, d2

    def gicpCovariances(self, k=20, epsilon=0.001):
        """GeneralizedIterativeClosestPoint::computeCovariances (impl/gicp.hpp:70-147) for the indexed cloud:
        (n, 3, 3) float64, NaN for points that are not in the index."""
        out = np.empty((self.n_cloud, 9), np.float64)
        check(self.lib.pclhip_gicp_covariances(self.h, int(k), float(epsilon), C.c_void_p(out.ctypes.data)), self.ctx.h)
        return out.reshape(self.n_cloud, 3, 3)

    def radiusSearch(self, queries, radius, max_nn=0):
        """Batch radiusSearch (search.h:271-273 / search.hpp:164-190).  Returns (offsets uint64 [nq+1],
        indices int32 [total], sqr_distances float32 [total]) -- neighbours of query i are
        indices[offsets[i]:offsets[i+1]], ascending by distance."""
        ptr, stride, nq, keep = _cloud(queries)
        offsets = np.zeros(nq + 1, np.uint64)
        total = C.c_uint64(0)
        optr = offsets.ctypes.data_as(C.POINTER(C.c_uint64))
        st = self.lib.pclhip_radius_search(self.h, ptr, stride, nq, float(radius), int(max_nn), optr, None, None, 0,
                                           C.byref(total))
        if st not in (0, -5):
            check(st, self.ctx.h)
        n = int(total.value)
        idx = np.empty(n, np.int32)
        d2 = np.empty(n, np.float32)
        if n:
            check(self.lib.pclhip_radius_search(self.h, ptr, stride, nq, float(radius), int(max_nn), optr,
                                                C.c_void_p(idx.ctypes.data), C.c_void_p(d2.ctypes.data), n,
                                                C.byref(total)), self.ctx.h)
        return offsets, idx, d2

    def setNormals(self, normals):
        ptr, stride, n, keep = _cloud(normals)
        assert n == self.n_cloud
        check(self.lib.pclhip_index_set_normals(self.h, ptr, stride), self.ctx.h)

    def _free(self):
        if getattr(self, "h", None):
            if self.ctx.h is not None:
                self.lib.pclhip_index_destroy(self.h)
            self.h = None
            self._cloud_id = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass


class NormalEstimation:
    """pcl::NormalEstimation<PointInT, pcl::Normal> with setKSearch (k-NN mode) or setRadiusSearch."""

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()
        self.lib = self.ctx.lib
        self.tree = None
        self.k = 0
        self.radius = 0.0
        self.vp = np.zeros(3, np.float32)  # sensor_origin_ default (normal_3d.h:328-351)
        self.cloud = None
        self.surface = None
        self.indices = None
        self.nan_count = 0

    def setInputCloud(self, cloud):
        self.cloud = cloud

    def setSearchSurface(self, cloud):
        """Feature::setSearchSurface (features/include/pcl/features/feature.h:139-153): neighbours are taken from this
        cloud, normals are computed at the input cloud's points (None: surface == input)."""
        self.surface = cloud

    def getSearchSurface(self):
        return self.surface

    def setIndices(self, indices):
        """PCLBase::setIndices (common/include/pcl/pcl_base.h:102-125): one normal per input[indices[j]]."""
        self.indices = None if indices is None else np.ascontiguousarray(indices, np.int32)

    def setSearchMethod(self, tree):
        self.tree = tree

    def getSearchMethod(self):
        return self.tree

    def getKSearch(self):
        return self.k

    def getRadiusSearch(self):
        return self.radius

    def getViewPoint(self):
        return tuple(float(v) for v in self.vp)

    def setKSearch(self, k):
        self.k = int(k)

    def setRadiusSearch(self, radius):
        self.radius = float(radius)

    def setViewPoint(self, x, y, z):
        self.vp = np.asarray([x, y, z], np.float32)

    def compute(self, want_output=True):
        """-> (n,4) float32 [nx,ny,nz,curvature]; also retains the normals inside the tree."""
        # Feature::initCompute (impl/feature.hpp:131-155): exactly one of k and radius must be set
        if self.radius != 0.0 and self.k != 0:
            raise ValueError("Both radius (%f) and K (%d) defined! Set one of them to zero first" % (self.radius, self.k))
        if self.radius == 0.0 and self.k == 0:
            raise ValueError("Neither radius nor K defined! Set one of them to a positive number first")
        if self.tree is None:
            self.tree = KdTree(self.ctx)
        if self.surface is not None or self.indices is not None:
            return self._compute_at(want_output)
        # feature.hpp:125-130 sets the search surface on the tree.  A tree the caller has already built
        # on this very cloud object is reused (setSearchMethod(tree) after tree.setInputCloud(cloud), the
        # common PCL idiom, would otherwise build twice).
        if self.tree.h is None or self.tree._cloud_id != (id(self.cloud), None):
            self.tree.setInputCloud(self.cloud)
        n = self.tree.n_cloud
        nan = C.c_uint64(0)
        out = None
        optr, stride = None, 0
        if want_output:
            stride = 16
            if _is_torch(self.cloud):
                import torch
                out = torch.empty((n, 4), dtype=torch.float32, device=self.cloud.device)
                optr = C.c_void_p(out.data_ptr())
            else:
                out = np.empty((n, 4), np.float32)
                optr = C.c_void_p(out.ctypes.data)
        if self.k > 0:
            check(self.lib.pclhip_normals(self.tree.h, self.k, _fp(self.vp), optr, stride, C.byref(nan)), self.ctx.h)
        else:
            check(self.lib.pclhip_normals_radius(self.tree.h, self.radius, _fp(self.vp), optr, stride, C.byref(nan)),
                  self.ctx.h)
        self.nan_count = int(nan.value)
        return out


    def _compute_at(self, want_output):
        """search surface != input and / or an index subset of the input (impl/feature.hpp:104-130,
        impl/normal_3d.hpp:48-95): the tree indexes the surface, every (selected) input point is a query."""
        surface = self.surface if self.surface is not None else self.cloud
        if self.tree.h is None or self.tree._cloud_id != (id(surface), None):
            self.tree.setInputCloud(surface)
        base, stride, nq, _keep = _cloud(self.cloud)
        m = nq if self.indices is None else len(self.indices)
        if _is_torch(self.cloud):
            import torch
            out = torch.empty((m, 4), dtype=torch.float32, device=self.cloud.device)
            optr = C.c_void_p(out.data_ptr())
        else:
            out = np.empty((m, 4), np.float32)
            optr = C.c_void_p(out.ctypes.data)
        nan = C.c_uint64(0)
        ind = None if self.indices is None else C.c_void_p(self.indices.ctypes.data)
        check(self.lib.pclhip_normals_at(self.tree.h, base, stride, nq, ind, 0 if self.indices is None else m, self.k,
                                         self.radius, _fp(self.vp), optr, 16, C.byref(nan)), self.ctx.h)
        self.nan_count = int(nan.value)
        return out if want_output else None


class CorrespondenceRejectorDistance:
    """pcl::registration::CorrespondenceRejectorDistance (correspondence_rejection_distance.h)."""
    kind = _lib.REJ_DISTANCE

    def __init__(self):
        self.param, self.min_corr = 0.0, 0

    def setMaximumDistance(self, d):
        self.param = float(d)

    def getMaximumDistance(self):
        return self.param


class CorrespondenceRejectorMedianDistance:
    """pcl::registration::CorrespondenceRejectorMedianDistance."""
    kind = _lib.REJ_MEDIAN_DISTANCE

    def __init__(self):
        self.param, self.min_corr = 1.0, 0
        self.median_distance_ = None

    def setMedianFactor(self, f):
        self.param = float(f)

    def getMedianFactor(self):
        return self.param

    def getMedianDistance(self):
        return self.median_distance_


class CorrespondenceRejectorOneToOne:
    """pcl::registration::CorrespondenceRejectorOneToOne."""
    kind = _lib.REJ_ONE_TO_ONE

    def __init__(self):
        self.param, self.min_corr = 0.0, 0


class CorrespondenceRejectorTrimmed:
    """pcl::registration::CorrespondenceRejectorTrimmed."""
    kind = _lib.REJ_TRIMMED

    def __init__(self):
        self.param, self.min_corr = 0.5, 0

    def setOverlapRatio(self, r):
        self.param = float(r)

    def getOverlapRatio(self):
        return self.param

    def setMinCorrespondences(self, n):
        self.min_corr = int(n)

    def getMinCorrespondences(self):
        return self.min_corr


def _set_filters(lib, ctx, h, rejectors, reciprocal):
    arr = (_lib.Rejector * max(len(rejectors), 1))()
    for i, r in enumerate(rejectors):
        arr[i].kind, arr[i].param, arr[i].min_correspondences = r.kind, r.param, r.min_corr
    check(lib.pclhip_icp_set_rejectors(h, arr, len(rejectors)), ctx.h)
    check(lib.pclhip_icp_set_reciprocal(h, int(bool(reciprocal))), ctx.h)


class CorrespondenceEstimation:
    """pcl::registration::CorrespondenceEstimation (determineCorrespondences only)."""

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()
        self.lib = self.ctx.lib
        self.tree = KdTree(self.ctx)
        self.icp = None
        self.src = None

    def setInputTarget(self, cloud):
        self._target = cloud
        self.tree.setInputCloud(cloud, getattr(self, "_tgt_indices", None))

    def setSearchMethodTarget(self, tree, force_no_recompute=False):
        self.tree = tree

    def setInputSource(self, cloud):
        self.src = cloud

    def setIndicesSource(self, indices):
        """correspondence_estimation.h:194: the source points to find correspondences for"""
        self._src_indices = None if indices is None else np.ascontiguousarray(indices, np.int32)

    def setIndicesTarget(self, indices):
        """correspondence_estimation.h:210: the target points correspondences may go to (the tree is rebuilt on
        them, impl/correspondence_estimation.hpp:82-91)"""
        self._tgt_indices = None if indices is None else np.ascontiguousarray(indices, np.int32)
        if getattr(self, "_target", None) is not None:
            self.tree.setInputCloud(self._target, self._tgt_indices)

    def determineReciprocalCorrespondences(self, max_distance=_SQRT_DBL_MAX):
        """impl/correspondence_estimation.hpp:220-311."""
        return self.determineCorrespondences(max_distance, reciprocal=True)

    def determineCorrespondences(self, max_distance=_SQRT_DBL_MAX, rejectors=(), reciprocal=False):
        """-> (index_query, index_match, distance) sorted by index_query (after `rejectors`: in the
        order the reference's chain leaves them)."""
        ptr, stride, n, keep = _cloud(self.src)
        h = C.c_void_p()
        check(self.lib.pclhip_icp_create(self.tree.h, C.byref(h)), self.ctx.h)
        try:
            ind = getattr(self, "_src_indices", None)
            if ind is None:
                check(self.lib.pclhip_icp_set_source(h, ptr, stride, n), self.ctx.h)
            else:
                check(self.lib.pclhip_icp_set_source_indexed(h, ptr, stride, n, C.c_void_p(ind.ctypes.data), len(ind)),
                      self.ctx.h)
            _set_filters(self.lib, self.ctx, h, list(rejectors), reciprocal)
            I = np.eye(4, dtype=np.float32).reshape(16)
            sums = np.zeros(_lib.NSUMS, np.float64)
            check(self.lib.pclhip_icp_iterate(h, _fp(I), float(max_distance), POINT_TO_POINT,
                                              sums.ctypes.data_as(C.POINTER(C.c_double))), self.ctx.h)
            q = np.empty(n, np.int32)
            m = np.empty(n, np.int32)
            d = np.empty(n, np.float32)
            cnt = C.c_uint64(0)
            check(self.lib.pclhip_icp_fetch_correspondences(
                h, C.c_void_p(q.ctypes.data), C.c_void_p(m.ctypes.data), C.c_void_p(d.ctypes.data),
                C.byref(cnt)), self.ctx.h)
            c = int(cnt.value)
            assert c == int(sums[28])
            for r in rejectors:
                if r.kind == _lib.REJ_MEDIAN_DISTANCE:
                    r.median_distance_ = float(self.lib.pclhip_icp_last_median_distance(h))
            return q[:c].copy(), m[:c].copy(), d[:c].copy()
        finally:
            self.lib.pclhip_icp_destroy(h)


class DefaultConvergenceCriteria:
    """pcl::registration::DefaultConvergenceCriteria (default_convergence_criteria.h:61-286) as a view of a
    registration's parameter block: hasConverged() itself runs on the device (closed_forms.hpp)."""

    def __init__(self, reg):
        self._reg = reg

    def setMaximumIterationsSimilarTransforms(self, n):
        self._reg.p.max_iterations_similar_transforms = int(n)

    def getMaximumIterationsSimilarTransforms(self):
        return int(self._reg.p.max_iterations_similar_transforms)

    def setFailureAfterMaximumIterations(self, f):
        self._reg.p.failure_after_max_iterations = int(bool(f))

    def getFailureAfterMaximumIterations(self):
        return bool(self._reg.p.failure_after_max_iterations)

    def setAbsoluteMSE(self, mse):
        self._reg.p.mse_threshold_absolute = float(mse)

    def getAbsoluteMSE(self):
        return float(self._reg.p.mse_threshold_absolute)

    # the remaining thresholds are overwritten from the registration's own setters at every align()
    # (impl/icp.hpp:157-161): the getters show what the loop will use
    def getMaximumIterations(self):
        return int(self._reg.p.max_iterations)

    def getRelativeMSE(self):
        return float(self._reg.p.euclidean_fitness_epsilon)

    def getTranslationThreshold(self):
        return float(self._reg.p.transformation_epsilon)

    def getRotationThreshold(self):
        e = float(self._reg.p.transformation_rotation_epsilon)
        return e if e > 0 else 0.99999

    def getConvergenceState(self):
        return self._reg.getConvergenceState()


class IterativeClosestPoint:
    """pcl::IterativeClosestPoint<PointXYZ, PointXYZ> (TransformationEstimationSVD)."""
    MODE = POINT_TO_POINT
    ORDER = 0

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()
        self.lib = self.ctx.lib
        self.p = IcpParams()
        self.lib.pclhip_icp_params_default(C.byref(self.p))
        self.p.mode = self.MODE
        self.tree = KdTree(self.ctx)
        self.h = None
        self.src = None
        self.src_normals = None
        self.target = None
        self._target_updated = False
        self._force_no_recompute = False
        self._src_dirty = True
        self._src_nrm_dirty = True
        self.result = None
        self._allreduce = None
        self.rejectors = []
        self.use_reciprocal = False
        self.ctx._adopt(self)

    def _release(self):
        self._drop_icp()

    # --- setters named after registration.h:276-415 ---
    def setInputTarget(self, cloud):
        """Registration::setInputTarget (impl/registration.hpp:53-66): remembers the cloud and flags it as
        updated; the tree is (re)built by the next align()/iterate() -- always, even for the same array
        object (it may have been modified in place), unless setSearchMethodTarget(tree, True) said not to."""
        self.target = cloud
        self._target_updated = True
        self._drop_icp()

    def setSearchMethodTarget(self, tree, force_no_recompute=False):
        """registration.h:214-221: with force_no_recompute the caller vouches that `tree` already indexes the
        target, and it is never rebuilt here."""
        self.tree = tree
        self._force_no_recompute = bool(force_no_recompute)
        if force_no_recompute:
            self._target_updated = False
        self._drop_icp()

    def setInputSource(self, cloud):
        """Registration::setInputSource (registration.h:195-196): every call marks the source as new (the
        device copy is refreshed by the next align()/iterate())."""
        self.src = cloud
        self._src_dirty = True
        self.src_normals = None   # a cloud without normals must not inherit the previous cloud's

    def setIndices(self, indices):
        """PCLBase::setIndices (common/include/pcl/pcl_base.h:102-125): only these source points take part in
        the correspondence search and the estimation; None = all points."""
        self._src_indices = None if indices is None else np.ascontiguousarray(indices, np.int32)
        self._src_dirty = True

    def setMaximumIterations(self, n):
        self.p.max_iterations = int(n)

    def setMaxCorrespondenceDistance(self, d):
        self.p.max_correspondence_distance = float(d)

    def setTransformationEpsilon(self, e):
        self.p.transformation_epsilon = float(e)

    def setTransformationRotationEpsilon(self, e):
        self.p.transformation_rotation_epsilon = float(e)

    def setEuclideanFitnessEpsilon(self, e):
        self.p.euclidean_fitness_epsilon = float(e)

    # --- getters of registration.h:190-330 ---
    def getClassName(self):
        return type(self).__name__

    def getInputSource(self):
        return self.src

    def getInputTarget(self):
        return self.target

    def getSearchMethodTarget(self):
        return self.tree

    def setSearchMethodSource(self, tree, force_no_recompute=False):
        """registration.h:230-250: the reciprocal search's tree over the source.  The device indexes the moving
        source itself every iteration; the object is kept for the getter."""
        self._tree_source = tree

    def getSearchMethodSource(self):
        return getattr(self, "_tree_source", None)

    def getMaximumIterations(self):
        return int(self.p.max_iterations)

    def getMaxCorrespondenceDistance(self):
        return float(self.p.max_correspondence_distance)

    def getTransformationEpsilon(self):
        return float(self.p.transformation_epsilon)

    def getTransformationRotationEpsilon(self):
        return float(self.p.transformation_rotation_epsilon)

    def getEuclideanFitnessEpsilon(self):
        return float(self.p.euclidean_fitness_epsilon)

    def setRANSACIterations(self, n):
        """registration.h:300-322: stored; IterativeClosestPoint does not use them (nor does the reference's)."""
        self._ransac_iterations = int(n)

    def getRANSACIterations(self):
        return getattr(self, "_ransac_iterations", 0)

    def setRANSACOutlierRejectionThreshold(self, t):
        self._inlier_threshold = float(t)

    def getRANSACOutlierRejectionThreshold(self):
        return getattr(self, "_inlier_threshold", 0.05)

    def getConvergeCriteria(self):
        """icp.h:180-184: the DefaultConvergenceCriteria options a caller reaches through the criteria object."""
        return DefaultConvergenceCriteria(self)

    def addCorrespondenceRejector(self, rejector):
        """Registration::addCorrespondenceRejector (registration.h:430-434)."""
        self.rejectors.append(rejector)
        self._filters_dirty = True

    def getCorrespondenceRejectors(self):
        return list(self.rejectors)

    def removeCorrespondenceRejector(self, i):
        """registration.h:536-544: False when there is no i-th rejector."""
        if i < 0 or i >= len(self.rejectors):
            return False
        del self.rejectors[i]
        self._filters_dirty = True
        return True

    def clearCorrespondenceRejectors(self):
        self.rejectors = []
        self._filters_dirty = True

    def getUseReciprocalCorrespondences(self):
        return self.use_reciprocal

    def setUseReciprocalCorrespondences(self, on):
        """icp.h:251-256."""
        self.use_reciprocal = bool(on)
        self._filters_dirty = True

    def setAllReduce(self, fn):
        """fn(device_ptr:int, count:int, stream:int) -> 0 ; sums the 32 doubles across ranks."""
        def tramp(user, ptr, count, stream):
            try:
                return int(fn(ptr, count, stream) or 0)
            except Exception:  # never raise through the C frame
                import traceback
                traceback.print_exc()
                return 1
        self._allreduce = _lib.ALLREDUCE_FN(tramp)
        if self.h:
            check(self.lib.pclhip_icp_set_allreduce(self.h, self._allreduce, None), self.ctx.h)

    def setCommunicator(self, comm):
        """Multi-GPU: sum the per-iteration record over the ranks of `comm` (a Communicator: RCCL over xGMI,
        issued from C on the context's stream).  None detaches."""
        self._comm = comm
        if self.h:
            check(self.lib.pclhip_icp_set_comm(self.h, comm.h if comm is not None else None), self.ctx.h)

    def setRegion(self, region):
        """Target sharding (pclhip_icp_set_region): only the source points whose CURRENT position lies in
        `region` = (lo.xyz, hi.xyz), half-open, take part on this rank.  None: all points."""
        self._region = None if region is None else np.ascontiguousarray(region, np.float32).reshape(6)
        if self.h:
            check(self.lib.pclhip_icp_set_region(self.h, _fp(self._region) if self._region is not None else None), self.ctx.h)

    def runSteps(self, n_steps, guess=None):
        """pclhip_icp_run_steps: exactly n_steps iterations queued back to back on the device, alignments
        restarting on convergence.  Returns the list of per-step dicts."""
        self._ensure()
        arr = (_lib.IcpStep * max(int(n_steps), 1))()
        g = None if guess is None else np.ascontiguousarray(guess, np.float32).reshape(16)
        check(self.lib.pclhip_icp_run_steps(self.h, C.byref(self.p), _fp(g) if g is not None else None, int(n_steps),
                                            arr), self.ctx.h)
        out = []
        for i in range(int(n_steps)):
            s = arr[i]
            out.append({"iteration": s.iteration, "state": _lib.CONVERGENCE_STATES[s.convergence_state],
                        "converged": bool(s.converged), "alignment_ended": bool(s.alignment_ended),
                        "num_correspondences": int(s.num_correspondences), "mse": float(s.mse),
                        "search_ms": float(s.search_ms), "kernels_ms": float(s.kernels_ms),
                        "step_ms": float(s.step_ms),
                        "final_transformation": np.array(s.final_transformation, np.float32).reshape(4, 4)})
        return out

    def _drop_icp(self):
        if getattr(self, "h", None):
            if self.ctx.h is not None:
                self.lib.pclhip_icp_destroy(self.h)
            self.h = None
        self._src_dirty = True
        self._src_nrm_dirty = True
        self._filters_dirty = True

    def _init_target(self):
        # Registration::initCompute (impl/registration.hpp:84-87)
        if self._target_updated and not self._force_no_recompute:
            self.tree.setInputCloud(self.target)
            self._after_target_build()
            self._target_updated = False
            self._drop_icp()
        if self.tree.h is None:
            raise ValueError("No input target dataset was given!")

    def _after_target_build(self):
        pass

    def _ensure(self):
        self._init_target()
        if self.src is None:
            raise ValueError("No input source dataset was given!")
        if self.h is None:
            h = C.c_void_p()
            check(self.lib.pclhip_icp_create(self.tree.h, C.byref(h)), self.ctx.h)
            self.h = h
            if self._allreduce is not None:
                check(self.lib.pclhip_icp_set_allreduce(self.h, self._allreduce, None), self.ctx.h)
            if getattr(self, "_comm", None) is not None:
                check(self.lib.pclhip_icp_set_comm(self.h, self._comm.h), self.ctx.h)
            if getattr(self, "_region", None) is not None:
                check(self.lib.pclhip_icp_set_region(self.h, _fp(self._region)), self.ctx.h)
        if self._src_dirty:
            ptr, stride, n, keep = _cloud(self.src)
            ind = getattr(self, "_src_indices", None)
            if ind is None:
                check(self.lib.pclhip_icp_set_source(self.h, ptr, stride, n), self.ctx.h)
            else:
                check(self.lib.pclhip_icp_set_source_indexed(self.h, ptr, stride, n, C.c_void_p(ind.ctypes.data), len(ind)),
                      self.ctx.h)
            self._src_dirty = False
            self._src_nrm_dirty = True
        if self.src_normals is not None and self._src_nrm_dirty:
            nptr, nstride, nn, _keep = _cloud(self.src_normals)
            assert nn == _cloud(self.src)[2], "one normal per source point"
            check(self.lib.pclhip_icp_set_source_normals(self.h, nptr, nstride), self.ctx.h)
        self._src_nrm_dirty = False
        self._apply_options()
        if getattr(self, "_filters_dirty", True):
            _set_filters(self.lib, self.ctx, self.h, self.rejectors, self.use_reciprocal)
            self._filters_dirty = False

    def _apply_options(self):
        pass

    def iterate(self, T_prev=None, max_dist=None):
        """One device iteration (search + accumulate kernels); returns the 32-double reduction record."""
        self._ensure()
        T = np.eye(4, dtype=np.float32) if T_prev is None else np.ascontiguousarray(T_prev, np.float32)
        sums = np.zeros(_lib.NSUMS, np.float64)
        md = self.p.max_correspondence_distance if max_dist is None else float(max_dist)
        check(self.lib.pclhip_icp_iterate(self.h, _fp(T.reshape(16)), md, self.MODE,
                                          sums.ctypes.data_as(C.POINTER(C.c_double))), self.ctx.h)
        return sums

    def reset(self):
        self._ensure()
        check(self.lib.pclhip_icp_reset(self.h), self.ctx.h)

    def lastKernelMs(self):
        """search + (filters) + accumulate kernels of the last iterate(), HIP events"""
        return float(self.lib.pclhip_icp_last_kernel_ms(self.h))

    def sourceOrderMs(self):
        """GPU time the last source upload spent on the spatial ordering (once per source cloud)"""
        self._ensure()
        return float(self.lib.pclhip_icp_source_order_ms(self.h))

    def lastSearchMs(self):
        """the search kernel alone"""
        return float(self.lib.pclhip_icp_last_search_ms(self.h))

    def solve(self, sums):
        T = np.zeros(16, np.float32)
        s = np.ascontiguousarray(sums, np.float64)
        check(self.lib.pclhip_solve_transformation(s.ctypes.data_as(C.POINTER(C.c_double)), self.MODE,
                                                   _fp(T)))
        return T.reshape(4, 4)

    def fetchCorrespondences(self):
        n = _cloud(self.src)[2]
        q = np.empty(n, np.int32)
        m = np.empty(n, np.int32)
        d = np.empty(n, np.float32)
        cnt = C.c_uint64(0)
        check(self.lib.pclhip_icp_fetch_correspondences(
            self.h, C.c_void_p(q.ctypes.data), C.c_void_p(m.ctypes.data), C.c_void_p(d.ctypes.data),
            C.byref(cnt)), self.ctx.h)
        c = int(cnt.value)
        return q[:c].copy(), m[:c].copy(), d[:c].copy()

    def align(self, guess=None, want_output=False):
        """Registration::align (registration.hpp:170-221).  Returns the registered source when
        want_output, else None; results via getFinalTransformation()/hasConverged()."""
        self._ensure()
        r = IcpResult()
        g = None
        if guess is not None:
            g = np.ascontiguousarray(guess, np.float32).reshape(16)
        check(self.lib.pclhip_icp_align(self.h, C.byref(self.p), _fp(g) if g is not None else None,
                                        C.byref(r)), self.ctx.h)
        self.result = r
        if want_output:
            return self.transformCloud(self.src, self.getFinalTransformation())
        return None

    def transformCloud(self, cloud, T):
        ptr, stride, n, keep = _cloud(cloud)
        Tm = np.ascontiguousarray(T, np.float32).reshape(16)
        if _is_torch(cloud):
            out = keep.clone()
            optr = C.c_void_p(out.data_ptr())
        else:
            out = keep.copy()
            optr = C.c_void_p(out.ctypes.data)
        check(self.lib.pclhip_transform_cloud(self.ctx.h, _fp(Tm), self.ORDER, ptr, optr, stride, n, 0),
              self.ctx.h)
        return out

    def getFinalTransformation(self):
        return np.array(self.result.final_transformation, np.float32).reshape(4, 4)

    def getFitnessScore(self, max_range=float(np.finfo(np.float64).max), transform=None):
        """Registration::getFitnessScore (impl/registration.hpp:132-168): mean squared 1-NN distance
        of the source moved by the final transformation (or `transform`), over the points whose
        squared distance is <= max_range; DBL_MAX when none qualifies."""
        self._ensure()
        T = self.getFinalTransformation() if transform is None else transform
        T = np.ascontiguousarray(T, np.float32).reshape(16)
        score = C.c_double(0.0)
        nr = C.c_uint64(0)
        check(self.lib.pclhip_icp_fitness_score(self.h, _fp(T), C.c_double(max_range), C.byref(score),
                                                C.byref(nr)), self.ctx.h)
        self.fitness_points = int(nr.value)
        return float(score.value)

    def getLastIncrementalTransformation(self):
        return np.array(self.result.last_transformation, np.float32).reshape(4, 4)

    def hasConverged(self):
        return bool(self.result.converged)

    def getConvergenceState(self):
        return _lib.CONVERGENCE_STATES[self.result.convergence_state]

    @property
    def nr_iterations_(self):
        return int(self.result.nr_iterations)

    def __del__(self):
        try:
            self._drop_icp()
        except Exception:
            pass


class IterativeClosestPointWithNormals(IterativeClosestPoint):
    """pcl::IterativeClosestPointWithNormals<PointNormal, PointNormal>
    (TransformationEstimationPointToPlaneLLS, icp.h:395-398).  Target normals come either from
    setInputTarget(cloud with >= 7 columns: x,y,z,_,nx,ny,nz) or from setTargetNormals()/a
    NormalEstimation run on the same KdTree."""
    MODE = POINT_TO_PLANE
    ORDER = 1

    def setInputTarget(self, cloud):
        super().setInputTarget(cloud)
        self._target_normals = None

    def _after_target_build(self):
        # pcl::PointNormal layout: normal at floats 4..6 (point_types.hpp:843-853)
        if self.target is not None and self.target.shape[1] >= 7:
            self.tree.setNormals(self.target[:, 4:7])
        if getattr(self, "_target_normals", None) is not None:
            self.tree.setNormals(self._target_normals)

    def setTargetNormals(self, normals):
        """Normals for the target given separately (one row per target point)."""
        self._target_normals = normals
        if self.tree.h is not None and not self._target_updated:
            self.tree.setNormals(normals)

    def setInputSource(self, cloud):
        super().setInputSource(cloud)
        if cloud.shape[1] >= 7:  # pcl::PointNormal source: its normals feed the symmetric objective
            self.src_normals = cloud[:, 4:7]
            self._src_nrm_dirty = True

    def setSourceNormals(self, normals):
        self.src_normals = normals
        self._src_nrm_dirty = True

    def setUseSymmetricObjective(self, on):
        """icp.h:380-400: TransformationEstimationSymmetricPointToPlaneLLS instead of PointToPlaneLLS."""
        self.MODE = _lib.SYMMETRIC if on else POINT_TO_PLANE
        self.p.mode = self.MODE

    def getUseSymmetricObjective(self):
        return self.MODE == _lib.SYMMETRIC

    def setEnforceSameDirectionNormals(self, on):
        """icp.h:416-428.  Only stored here (this is usually called before the clouds are set); applied
        whenever the device-side registration object is (re)created."""
        self._enforce = bool(on)

    def _apply_options(self):
        check(self.lib.pclhip_icp_set_enforce_same_direction_normals(
            self.h, 1 if getattr(self, "_enforce", True) else 0), self.ctx.h)

    def getEnforceSameDirectionNormals(self):
        return getattr(self, "_enforce", True)


def estimateRigidTransformation(ctx, mode, src, tgt, src_normals=None, tgt_normals=None,
                                enforce_same_direction_normals=True, weights=None):
    """TransformationEstimation{SVD, PointToPlaneLLS, SymmetricPointToPlaneLLS}::estimateRigidTransformation
    (cloud_src, cloud_tgt) for equally sized clouds (pair i = (src[i], tgt[i])).  Returns (T 4x4, sums)."""
    sp, ss, n, _k1 = _cloud(src)
    tp, ts, nt, _k2 = _cloud(tgt)
    assert n == nt, "Number or points in source differs than target"
    snp, sns = None, 0
    tnp, tns = None, 0
    if src_normals is not None:
        snp, sns, _, _k3 = _cloud(src_normals)
    if tgt_normals is not None:
        tnp, tns, _, _k4 = _cloud(tgt_normals)
    T = np.zeros(16, np.float32)
    sums = np.zeros(_lib.NSUMS, np.float64)
    if weights is not None:  # TransformationEstimationPointToPlaneLLSWeighted::setWeights
        assert int(mode) == POINT_TO_PLANE, "weights belong to the point-to-plane estimator"
        w = np.ascontiguousarray(weights, np.float32)
        assert w.shape == (n,), "Number or weights from the number of correspondences"
        check(ctx.lib.pclhip_estimate_rigid_transformation_weighted(
            ctx.h, sp, ss, tp, ts, tnp, tns, C.c_void_p(w.ctypes.data), n, _fp(T),
            sums.ctypes.data_as(C.POINTER(C.c_double))), ctx.h)
        return T.reshape(4, 4), sums
    check(ctx.lib.pclhip_estimate_rigid_transformation(
        ctx.h, int(mode), sp, ss, snp, sns, tp, ts, tnp, tns, n, 1 if enforce_same_direction_normals else 0,
        _fp(T), sums.ctypes.data_as(C.POINTER(C.c_double))), ctx.h)
    return T.reshape(4, 4), sums


def getPCDHeader(path):
    """PCDReader::readHeader (io/src/pcd_io.cpp:115-392) -> _lib.PcdInfo"""
    info = _lib.PcdInfo()
    check(_lib.load().pclhip_pcd_read_header(os.fsencode(path), C.byref(info)))
    return info


def loadPCDFile(path, with_normals=False, device=None):
    """pcl::io::loadPCDFile into PointXYZ records [n,4] (x,y,z,1) or PointNormal records [n,12]
    (x,y,z,1, nx,ny,nz,0, curvature,0,0,0).  device=None -> numpy array, else a torch tensor on that
    device (the library writes straight into its memory).  Returns (cloud, is_dense)."""
    lib = _lib.load()
    info = getPCDHeader(path)
    n = int(info.points)
    cols = 12 if with_normals else 4
    dense = C.c_int(1)
    cnt = C.c_uint64(0)
    if device is None:
        out = np.zeros((n, cols), np.float32)
        ptr = C.c_void_p(out.ctypes.data)
    else:
        import torch
        out = torch.zeros((n, cols), dtype=torch.float32, device=device)
        ptr = C.c_void_p(out.data_ptr())
    check(lib.pclhip_pcd_read(os.fsencode(path), ptr, cols * 4, 16 if with_normals else 0, n, C.byref(cnt),
                              C.byref(dense)))
    return out, bool(dense.value)


def loadPCDField(path, field, component=0):
    """One component of any field of a PCD file as float32 [n] (packed rgb/rgba: view the result as uint32)."""
    info = getPCDHeader(path)
    out = np.zeros(int(info.points), np.float32)
    cnt = C.c_uint64(0)
    check(_lib.load().pclhip_pcd_read_field(os.fsencode(path), field.encode(), int(component),
                                            C.c_void_p(out.ctypes.data), len(out), C.byref(cnt)))
    return out


def savePCDFile(path, cloud, mode="binary", precision=8, width=None, height=None, viewpoint=None):
    """pcl::io::savePCDFile{ASCII,Binary,BinaryCompressed}; clouds with >= 7 columns (PointNormal layout:
    normals at floats 4..6, curvature at float 8 when present) are written with their normals."""
    data_type = {"ascii": 0, "binary": 1, "binary_compressed": 2}[mode]
    ptr, stride, n, keep = _cloud(cloud)
    ncol = stride // 4
    if width is None and height is None and viewpoint is None:
        check(_lib.load().pclhip_pcd_write(os.fsencode(path), ptr, stride, 16 if ncol >= 7 else 0, n, data_type,
                                           int(precision)))
        return
    width = n if width is None else int(width)
    height = 1 if height is None else int(height)
    assert width * height == n, "width x height must equal the number of points"
    vp = None
    if viewpoint is not None:
        vp = np.ascontiguousarray(viewpoint, np.float32)
        assert vp.shape == (7,)
    check(_lib.load().pclhip_pcd_write_organized(os.fsencode(path), ptr, stride, 16 if ncol >= 7 else 0, width, height,
                                                 _fp(vp) if vp is not None else None, data_type, int(precision)))


class VoxelGrid:
    """pcl::VoxelGrid<PointT> for pcl::PointXYZ ((n, 4) clouds) and pcl::PointNormal ((n, 12) clouds): leaf size,
    minimum points per voxel, downsample_all_data, the pass-through filter on one field, the leaf layout."""

    _FLT_MAX = float(np.finfo(np.float32).max)
    # position of a field inside the record, in floats (point_types.hpp:315-321, 843-853)
    _FIELDS = {"x": 0, "y": 1, "z": 2, "normal_x": 4, "normal_y": 5, "normal_z": 6, "curvature": 8}

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()
        self.lib = self.ctx.lib
        self.leaf = np.zeros(3, np.float32)
        self.min_pts = 0
        self.field = ""                                  # voxel_grid.h:503-512: no filter until a field is named
        self.limits = (-self._FLT_MAX, self._FLT_MAX)
        self.negative = False
        self.cloud = None

    def setInputCloud(self, cloud):
        self.cloud = cloud

    def setLeafSize(self, lx, ly=None, lz=None):
        self.leaf = np.asarray([lx, lx if ly is None else ly, lx if lz is None else lz], np.float32)

    def setMinimumPointsNumberPerVoxel(self, n):
        self.min_pts = int(n)

    def getLeafSize(self):
        return tuple(float(v) for v in self.leaf)

    def getMinimumPointsNumberPerVoxel(self):
        return self.min_pts

    def setFilterFieldName(self, name):
        """voxel_grid.h:440-444: points are filtered on this field before the grid is laid out ("" = no filter)"""
        self.field = str(name)

    def getFilterFieldName(self):
        return self.field

    def setFilterLimits(self, lo, hi):
        self.limits = (float(lo), float(hi))

    def getFilterLimits(self):
        return self.limits

    def setFilterLimitsNegative(self, negative):
        """voxel_grid.h:468-476: True keeps the points OUTSIDE (limit_min, limit_max) instead of those inside"""
        self.negative = bool(negative)

    def getFilterLimitsNegative(self):
        return self.negative

    def _limit_flags(self, stride):
        if not self.field:
            return 0
        idx = self._FIELDS.get(self.field)
        if idx is None or (idx + 1) * 4 > stride or (idx > 2 and stride < 48):
            raise ValueError("[pcl::VoxelGrid] could not find field '%s' in this point type" % self.field)
        return 1 | (2 if self.negative else 0) | ((idx + 1) << 8)

    def setDownsampleAllData(self, downsample):
        """voxel_grid.h:293-302: False averages only x, y, z and leaves the other fields of the output at 0"""
        self.all_data = bool(downsample)

    def getDownsampleAllData(self):
        return getattr(self, "all_data", True)

    def filter(self):
        """-> [m, 4] (x, y, z, 1) for PointXYZ clouds; clouds with >= 12 columns are pcl::PointNormal records
        (x y z 1 | nx ny nz 0 | curvature 0 0 0) and come back in the same layout."""
        ptr, stride, n, keep = _cloud(self.cloud)
        cnt = C.c_uint64(0)
        has = self._limit_flags(stride)
        lo, hi = self.limits
        with_normals = stride >= 48
        cols = 12 if with_normals else 4
        if _is_torch(self.cloud):
            import torch
            out = torch.empty((max(n, 1), cols), dtype=torch.float32, device=self.cloud.device)
            optr = C.c_void_p(out.data_ptr())
        else:
            out = np.empty((max(n, 1), cols), np.float32)
            optr = C.c_void_p(out.ctypes.data)
        from ._lib import VoxelGridDims
        dims = VoxelGridDims()
        layout = None
        if getattr(self, "save_leaf_layout", False) and n > 0:
            # the layout has one int per cell of the grid: ask for the grid first (a bounding-box pass)
            check(self.lib.pclhip_voxelgrid_grid(self.ctx.h, ptr, stride, n, _fp(self.leaf), int(has), lo, hi,
                                                 C.byref(dims)), self.ctx.h)
            layout = np.empty(int(dims.div_b[0]) * int(dims.div_b[1]) * int(dims.div_b[2]), np.int32)
        check(self.lib.pclhip_voxelgrid_ex2(self.ctx.h, ptr, stride, n, _fp(self.leaf), self.min_pts, int(has), lo, hi,
                                            int(self.getDownsampleAllData()), 16 if with_normals else 0, optr, cols * 4,
                                            C.byref(cnt), None if layout is None else C.c_void_p(layout.ctypes.data),
                                            0 if layout is None else len(layout), C.byref(dims)), self.ctx.h)
        self._dims = dims
        self._layout = layout
        return out[:int(cnt.value)]

    # ---- the grid of the last filter() call and the leaf layout (voxel_grid.h:316-420) ----
    def setSaveLeafLayout(self, save):
        self.save_leaf_layout = bool(save)

    def getSaveLeafLayout(self):
        return getattr(self, "save_leaf_layout", False)

    def getMinBoxCoordinates(self):
        return np.asarray(self._dims.min_b, np.int32)

    def getMaxBoxCoordinates(self):
        return np.asarray(self._dims.max_b, np.int32)

    def getNrDivisions(self):
        return np.asarray(self._dims.div_b, np.int32)

    def getDivisionMultiplier(self):
        return np.asarray(self._dims.divb_mul, np.int32)

    def getLeafLayout(self):
        """position (i-min_x) + (j-min_y)*div_x + (k-min_z)*div_x*div_y -> index of that voxel's centroid, -1 if empty"""
        return np.empty(0, np.int32) if self._layout is None else self._layout

    def getGridCoordinates(self, x, y, z):
        inv = np.float32(1.0) / self.leaf
        return np.floor(np.asarray([x, y, z], np.float32) * inv).astype(np.int32)      # voxel_grid.h:402-407

    def getCentroidIndexAt(self, ijk):
        idx = int(np.dot(np.asarray(ijk, np.int64) - self.getMinBoxCoordinates(), self.getDivisionMultiplier()))
        if self._layout is None or idx < 0 or idx >= len(self._layout):
            return -1                                                                   # voxel_grid.h:412-421
        return int(self._layout[idx])

    def getCentroidIndex(self, p):
        return self.getCentroidIndexAt(self.getGridCoordinates(p[0], p[1], p[2]))
