/*
 * pclhip.h -- C ABI of the MI355X-native ICP hot path (k-NN correspondence search, normal
 * estimation, point-to-point / point-to-plane ICP, VoxelGrid prefilter).
 *
 * PCL has no FFI: the path sits behind C++ virtual plugin points.  Each entry point below names
 * the reference interface it replaces (file:line relative to the PCL tree); the adapter classes
 * in include/pclhip/pcl_compat.hpp (and the real-PCL subclasses shown in INTEGRATION.md) are the
 * only intended callers.
 *
 * Conventions
 *  - plain C, opaque handles, no exceptions across the boundary; every function returns
 *    PCLHIP_OK (0) or a negative pclhip_status; pclhip_last_error() gives the message.
 *  - point buffers are arrays of records with a byte stride; x,y,z are three consecutive floats
 *    at byte 0 of each record (pcl::PointXYZ = 16 B, pcl::PointNormal = 48 B with the normal at
 *    +16, common/include/pcl/impl/point_types.hpp:315-321,843-853).  Buffers may live in host OR
 *    device memory -- the library detects which (hipPointerGetAttributes) and stages host data.
 *    A DEVICE buffer must be complete when it is handed over: the context works on its own stream and
 *    does not wait for the stream that is still producing the buffer (create the context on that stream
 *    with pclhip_ctx_create_on_stream, or synchronise the producer first).
 *  - index_t is int32 (common/include/pcl/types.h:110-133); "no neighbour" is -1, distance +inf.
 *  - a handle is not thread-safe, with one exception: the QUERY entry points pclhip_knn and
 *    pclhip_radius_search may be called concurrently on one index / context (PCL calls its `const`
 *    search virtuals from OpenMP loops: registration/.../impl/correspondence_estimation.hpp:163-175,
 *    features/.../impl/normal_3d_omp.hpp:76-81, search/.../impl/search.hpp:164-190); such callers are
 *    served one after the other (a lock of the context), not in parallel.  Building, destroying or
 *    aligning on the same context meanwhile is the caller's race.  Distinct contexts may be used
 *    concurrently.  All work of a context is issued on its HIP stream; entry points that return host
 *    results synchronise it.
 *  - results: k-NN indices/distances are bit-exact w.r.t. the CPU oracle (float L2_Simple
 *    ((dx*dx)+dy*dy)+dz*dz, ascending (distance, index), ties -> lower index).
 */
#ifndef PCLHIP_H_
#define PCLHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define PCLHIP_API __attribute__((visibility("default")))
#else
#define PCLHIP_API
#endif

typedef int pclhip_status;
enum {
  PCLHIP_OK = 0,
  PCLHIP_ERR_INVALID = -1,  /* bad argument */
  PCLHIP_ERR_HIP = -2,      /* HIP runtime failure (message has hipGetErrorString) */
  PCLHIP_ERR_NO_DEVICE = -3,
  PCLHIP_ERR_STATE = -4,    /* call order (e.g. point-to-plane without target normals) */
  PCLHIP_ERR_OVERFLOW = -5  /* VoxelGrid: leaf too small, int32 voxel index would overflow */
};

typedef struct pclhip_ctx pclhip_ctx;
typedef struct pclhip_index pclhip_index;
typedef struct pclhip_icp pclhip_icp;

/* ---- context --------------------------------------------------------------------------------
 * stream: a hipStream_t to issue work on (e.g. torch.cuda.current_stream().cuda_stream), or NULL
 * to let the context create its own non-blocking stream. */
PCLHIP_API pclhip_status pclhip_ctx_create(int device, void* stream, pclhip_ctx** out);
/* Same, but `stream` is adopted as given -- including NULL, which then means the device's legacy default
 * stream (what torch.cuda.current_stream().cuda_stream is when torch runs on its default stream), not
 * "create one".  Use this whenever the caller orders its own work (collectives, copies) on that stream. */
PCLHIP_API pclhip_status pclhip_ctx_create_on_stream(int device, void* stream, pclhip_ctx** out);
/* the hipStream_t the context issues its work on */
PCLHIP_API void* pclhip_ctx_stream(const pclhip_ctx* ctx);
PCLHIP_API void pclhip_ctx_destroy(pclhip_ctx* ctx);
PCLHIP_API const char* pclhip_last_error(const pclhip_ctx* ctx /* may be NULL */);
PCLHIP_API pclhip_status pclhip_ctx_synchronize(pclhip_ctx* ctx);
/* Reserve `bytes` of device memory as the context's arena: index arrays, registration state and every temporary are
 * carved out of it (what does not fit falls back to hipMalloc), so the first index build of a context costs what a
 * rebuild costs.  Without this call the context reserves 288 bytes per point of the first cloud of a million points or
 * more that it sees (option "arena_mb" overrides; 0 = no arena).  No PCL counterpart: PCL's containers allocate on the
 * host (common/include/pcl/point_cloud.h:393-409); this is the device side of that. */
PCLHIP_API pclhip_status pclhip_ctx_reserve(pclhip_ctx* ctx, uint64_t bytes);
/* The tuning knobs of a context, in one place (the library reads no environment variable; none of these changes a result):
 *   "served_groups"  0 | 1  target sharding: pclhip_icp_align / pclhip_icp_run_steps walk only the 64-point groups the rank
 *                           serves (default 1; 0 = every launch walks the whole source -- the reference of the tests)
 *   "icp_lookahead"  n      pclhip_icp_align: iterations queued ahead of the host's knowledge (default 1)
 *   "cache_mb"       n      device blocks kept for reuse between calls, MiB (default 16384, at most a quarter of the device)
 *   "arena_mb"       n      size of the automatic arena (pclhip_ctx_reserve), MiB; 0 = none (default: 288 B per point);
 *                           PCLHIP_ERR_STATE once the arena exists
 *   "cell_start"     0 | 1  seeded descents test the kd CELL of the level-2 / level-3 node of a seed (pclhip_index_cells)
 *                           instead of its tight box when they look for a start level (default 1)
 *   "reseed"         0 | 1  a 64-query group whose seeds are all farther than two leaf diagonals gets one fresh seed from a
 *                           greedy walk down the hierarchy (default 1)
 *   "lane_search"    0 | 1  the seeded launches of an ICP iteration run one lane per query over the quad levels and their
 *                           cells (lane.hip) instead of the wave-cooperative traversal (default 0: exact, measured 2.5x
 *                           slower at 10M points -- and in the device-driven loop every iteration then queues the
 *                           fall-through search kernel plus three lane kernels, launch overhead that figure does not
 *                           separate out); "lane_max_up" n (default 2) quad levels the first pass climbs,
 *                           "lane_far" x (default 0.25) squared mean leaf diagonals beyond which a seed is replaced
 *   "standoff_thickness" x, "standoff_max_mb" n   which indices the launch that starts an alignment searches by the stand-off
 *                           body (leaf discs): leaves thinner than x against their width (default 0.3) and an index of at
 *                           most n MiB (default: no limit); everything else goes through the seeded body with disc bounds
 * No PCL counterpart (PCL's knobs are the setters of its classes, which the bindings map onto the calls below). */
PCLHIP_API pclhip_status pclhip_ctx_set_option(pclhip_ctx* ctx, const char* name, double value);
PCLHIP_API const char* pclhip_version(void);
/* Optional traversal work counters (diagnostics): enable != 0 allocates/zeroes 8 device counters
 * that every search kernel of this context adds to; out (8 x uint64, may be NULL) receives the
 * current values: [0] interior nodes scanned, [1] leaves past the group test, [2] leaves past the
 * per-lane test (16-candidate all-pairs blocks), [3] stack pushes, [4] 64-query groups.  (With "lane_search" the lane
 * kernels count in the same slots: [0] queries, [1] done with their own leaf, [2] done in the first pass, [3] greedy
 * descents, [4] finished by the second pass.) */
PCLHIP_API pclhip_status pclhip_ctx_stats(pclhip_ctx* ctx, int enable, uint64_t* out);

/* ---- spatial index over the target cloud ------------------------------------------------------
 * Replaces pcl::KdTreeFLANN<PointT>::setInputCloud (kdtree/include/pcl/kdtree/impl/
 * kdtree_flann.hpp:99-136) as called from pcl::search::KdTree<PointT>::setInputCloud
 * (search/include/pcl/search/impl/kdtree.hpp:87-97) and Registration::initCompute
 * (registration/include/pcl/registration/impl/registration.hpp:84-87).
 * Non-finite points are dropped (kdtree_flann.hpp:443-452).  `indices` (optional) selects a subset
 * exactly like the (cloud, indices) overload (:428-498): returned neighbour indices are indices
 * into the ORIGINAL cloud. */
PCLHIP_API pclhip_status pclhip_index_build(pclhip_ctx* ctx, const void* points, size_t stride_bytes,
                                            uint64_t n, const int32_t* indices, uint64_t n_indices,
                                            pclhip_index** out);
/* The same through a pcl::PointRepresentation that maps a point to (x, y, z) times per-axis rescale values
 * (pcl::search::KdTree::setPointRepresentation, search/include/pcl/search/kdtree.h:110; KdTreeFLANN vectorises
 * every point with it, kdtree_flann.hpp:463-498; CustomPointRepresentation / setRescaleValues,
 * common/include/pcl/point_representation.h:150-190,546-579): the index holds coordinate * scale[axis], queries of
 * pclhip_knn / pclhip_radius_search are mapped the same way, distances are those of the rescaled space.  A
 * scale of 0 removes the axis (e.g. {1, 1, 0}: the x-y representation; that coordinate need not be finite).
 * scale == NULL: the default representation.  Registration (pclhip_icp_create) needs the default one. */
PCLHIP_API pclhip_status pclhip_index_build_scaled(pclhip_ctx* ctx, const void* points, size_t stride_bytes,
                                                   uint64_t n, const int32_t* indices, uint64_t n_indices,
                                                   const float scale[3], pclhip_index** out);
/* The same, and the records' own normals -- (nx, ny, nz, curvature) at `normals_offset_bytes` inside every record, 0 =
 * none -- attached to the index from the same upload (pclhip_index_set_normals would stage the cloud a second time):
 * pcl::PointNormal clouds as IterativeClosestPointWithNormals takes them (registration/include/pcl/registration/icp.h
 * :330-345; the normals sit at +16). */
PCLHIP_API pclhip_status pclhip_index_build_ex(pclhip_ctx* ctx, const void* points, size_t stride_bytes, uint64_t n,
                                               const int32_t* indices, uint64_t n_indices, const float* scale,
                                               size_t normals_offset_bytes, pclhip_index** out);
PCLHIP_API void pclhip_index_destroy(pclhip_index* index);
/* number of finite points indexed */
PCLHIP_API uint64_t pclhip_index_size(const pclhip_index* index);
/* milliseconds of GPU time spent in the last build (bbox + kd ordering by radix-sort rounds + gather + boxes) */
PCLHIP_API double pclhip_index_build_ms(const pclhip_index* index);
/* The order the index keeps its points in: out[j] = original index of the point at position j (pclhip_index_size entries,
 * host or device memory).  Consecutive positions are spatial neighbours -- every aligned run of 16 * 4^k positions is one
 * cell of a kd partition -- so a caller can lay per-point attributes out the same way.  (KdTreeFLANN keeps the
 * corresponding permutation private: kdtree_flann.h `index_mapping_`; tests use this to check the cells.) */
PCLHIP_API pclhip_status pclhip_index_order(pclhip_index* index, int32_t* out);
/* The structure the per-lane search of the seeded ICP launches walks (round 5): the kd order seen as a 4-ary tree over
 * the 16-point leaves.  Level q has ceil(leaves / 4^q) nodes; node i covers positions [16 i 4^q, 16 (i+1) 4^q) of
 * pclhip_index_order.  For one level: boxes[6 i ..] = tight AABB (lo xyz, hi xyz) of node i, cells[6 i ..] = its CELL --
 * an axis-aligned region, +-inf where unbounded, whose INTERIOR holds no point of any other node (what lets a search
 * stop inside the node; an inverted cell, lo > hi, marks a node whose siblings the build could not separate).
 * *count receives the number of nodes of the level, *top_level the root's level; buffers (host memory, `capacity` nodes
 * each, either may be NULL) are filled when capacity suffices.  No counterpart in PCL (FLANN keeps its tree private);
 * tests check the cells against the points with it.  PCLHIP_ERR_STATE for an index without the structure. */
PCLHIP_API pclhip_status pclhip_index_cells(pclhip_index* index, int level, float* boxes, float* cells, uint64_t capacity,
                                            uint64_t* count, int* top_level);

/* Exact k nearest neighbours of nq query points.
 * Replaces pcl::KdTreeFLANN<PointT>::nearestKSearch (kdtree_flann.hpp:234-274) and the batch
 * overloads of pcl::search::Search<PointT>::nearestKSearch (search/include/pcl/search/impl/
 * search.hpp:113-136).  k is clamped to the index size (:241-242); unused slots get -1 / +inf.
 * out_idx: nq*k int32, out_d2: nq*k float (squared L2), both host or device memory.
 * Non-finite queries return no neighbours. */
PCLHIP_API pclhip_status pclhip_knn(pclhip_index* index, const void* queries, size_t stride_bytes,
                                    uint64_t nq, int k, int32_t* out_idx, float* out_d2);

/* All neighbours within `radius` of nq query points, as a CSR list.
 * Replaces pcl::KdTreeFLANN<PointT>::radiusSearch (kdtree_flann.hpp:372-414) and the batch overload of
 * pcl::search::Search<PointT>::radiusSearch (search/include/pcl/search/impl/search.hpp:164-190):
 * squared distance < float(radius*radius), ascending (distance, index); max_nn > 0 keeps only the
 * max_nn nearest of each query (max_nn = 0: all).  out_offsets: nq+1 uint64 in HOST memory (always
 * written); *out_total = out_offsets[nq].  out_idx/out_d2 (host or device) must hold `capacity`
 * entries; if capacity < *out_total the call returns PCLHIP_ERR_OVERFLOW after writing the offsets,
 * so a caller sizes the buffers with a first call (capacity 0) and fetches with a second. */
PCLHIP_API pclhip_status pclhip_radius_search(pclhip_index* index, const void* queries, size_t stride_bytes,
                                              uint64_t nq, double radius, uint32_t max_nn, uint64_t* out_offsets,
                                              int32_t* out_idx, float* out_d2, uint64_t capacity,
                                              uint64_t* out_total);

/* Surface normals + curvature of every indexed point from its k nearest neighbours in the index.
 * Replaces pcl::NormalEstimation<PointInT,PointOutT>::computeFeature (features/include/pcl/
 * features/impl/normal_3d.hpp:48-95) with setKSearch(k), search surface == input:
 * computeMeanAndCovarianceMatrix (common/include/pcl/common/impl/centroid.hpp:581-650) ->
 * solvePlaneParameters/eigen33 (features/include/pcl/features/impl/feature.hpp:64-89,
 * common/include/pcl/common/impl/eigen.hpp:295-325) -> flipNormalTowardsViewpoint
 * (features/include/pcl/features/normal_3d.h:169-188).
 * out (optional, host or device): one record per ORIGINAL cloud point (n of pclhip_index_build):
 * 4 floats nx,ny,nz,curvature at byte 0 of each out_stride_bytes record; NaN for dropped points.
 * The normals are also retained inside the index for point-to-plane ICP.  out_nan_count optional. */
PCLHIP_API pclhip_status pclhip_normals(pclhip_index* index, int k, const float viewpoint[3],
                                        void* out, size_t out_stride_bytes, uint64_t* out_nan_count);
/* The same normals (k >= 1 and radius 0, or k 0 and radius > 0) written as WHOLE output records, the way
 * Feature::compute leaves a PointCloud<pcl::Normal> (impl/feature.hpp:195-229: the cloud is resized -- value-initialised
 * records -- and computeFeature fills normal[0..2] and curvature, normal_3d.hpp:60-66): every record_bytes record is
 * zeroed, the normal goes to normal_offset, the curvature to curvature_offset (pcl::Normal: 32 / 0 / 16).  A host `out`
 * costs one linear copy instead of 16-byte rows into a staging array plus the caller's unpacking loop: a repeated
 * NormalEstimation::compute of 10M points through the binding 87 -> 7.6 ms, a first one 185 -> 94 ms (of which 60 are
 * PCL's own resize of the output cloud; scratch/boundary_probe.cpp). */
PCLHIP_API pclhip_status pclhip_normals_records(pclhip_index* index, int k, double radius, const float viewpoint[3],
                                                void* out, size_t record_bytes, size_t normal_offset,
                                                size_t curvature_offset, uint64_t* out_nan_count);
/* Same with setRadiusSearch(radius) (Feature::compute, features/include/pcl/features/impl/feature.hpp:140-155):
 * the plane is fitted to ALL indexed points with squared distance < float(radius^2), taken in the
 * order radiusSearch returns them (ascending distance); fewer than 3 neighbours -> NaN. */
PCLHIP_API pclhip_status pclhip_normals_radius(pclhip_index* index, double radius, const float viewpoint[3],
                                               void* out, size_t out_stride_bytes, uint64_t* out_nan_count);
/* The same two searches with a SEARCH SURFACE different from the input cloud, and/or over an index subset of the
 * input: Feature::setSearchSurface / PCLBase::setIndices (features/include/pcl/features/feature.h:139-153,
 * impl/feature.hpp:104-118; common/include/pcl/pcl_base.h:102-125) as NormalEstimation::computeFeature uses them
 * (impl/normal_3d.hpp:48-95): `index` holds the surface; query j is record indices[j] of `queries` (record j when
 * indices is NULL); its plane is fitted to the surface points the search returns, in that order, and the normal is
 * flipped towards the viewpoint as seen from the QUERY point.  Exactly one of k >= 1 / radius > 0.
 * out (host or device): nx,ny,nz,curvature at byte 0 of out_stride_bytes records, one per query; NaN where the query
 * is non-finite or has fewer than 3 neighbours.  Nothing is retained in the index.
 * Radius mode accumulates the covariance inside the traversal, shifted by the QUERY point and summed in double in
 * traversal order; the reference shifts by the first neighbour returned and sums in float in ascending-distance order
 * (common/include/pcl/common/impl/centroid.hpp:581-650).  For self-queries the two shifts coincide; for a search surface
 * that is not the input the results meet the 1e-5 contract on normals and curvature, not bit parity. */
PCLHIP_API pclhip_status pclhip_normals_at(pclhip_index* surface, const void* queries, size_t stride_bytes, uint64_t n_queries,
                                           const int32_t* indices, uint64_t n_indices, int k, double radius,
                                           const float viewpoint[3], void* out, size_t out_stride_bytes,
                                           uint64_t* out_nan_count);
/* GeneralizedIterativeClosestPoint::computeCovariances (registration/include/pcl/registration/impl/gicp.hpp
 * :70-147): for every indexed point the covariance of its k nearest neighbours (k_correspondences_, default
 * 20, <= 32 here), regularised to singular values (1, 1, epsilon) (gicp_epsilon_, default 0.001).
 * out (host or device): 9 doubles (row-major 3x3) per ORIGINAL cloud point; NaN for dropped points. */
PCLHIP_API pclhip_status pclhip_gicp_covariances(pclhip_index* index, int k, double epsilon, double* out);
/* GPU time (ms) of the last pclhip_knn / pclhip_normals traversal kernel on this index. */
PCLHIP_API double pclhip_index_last_kernel_ms(const pclhip_index* index);
/* Supply target normals computed elsewhere (e.g. a pcl::PointNormal target: normals = points + 16,
 * stride 48).  One record per ORIGINAL cloud point. */
PCLHIP_API pclhip_status pclhip_index_set_normals(pclhip_index* index, const void* normals,
                                                  size_t stride_bytes);

/* ---- ICP --------------------------------------------------------------------------------------*/
enum { PCLHIP_ICP_POINT_TO_POINT = 0, /* TransformationEstimationSVD, icp.h:149-151 */
       PCLHIP_ICP_POINT_TO_PLANE = 1, /* TransformationEstimationPointToPlaneLLS, icp.h:395-398 */
       PCLHIP_ICP_SYMMETRIC = 2       /* TransformationEstimationSymmetricPointToPlaneLLS,
                                         IterativeClosestPointWithNormals::setUseSymmetricObjective, icp.h:380-400;
                                         needs source normals too (pclhip_icp_set_source_normals) */ };

/* number of doubles in the per-iteration reduction record */
#define PCLHIP_ICP_NSUMS 32
/* layout of sums[]:
 *   point-to-plane: [0..20] upper triangle of ATA (row-major, order of impl/
 *                   transformation_estimation_point_to_plane_lls.hpp:213-233), [21..26] ATb
 *   symmetric:      same slots, for v = [(p+q) x n ; n] and rhs v ((q-p).n)
 *                   (impl/transformation_estimation_symmetric_point_to_plane_lls.hpp:161-190)
 *   point-to-point: [0..2] sum s, [3..5] sum t, [6..14] sum t_i*s_j (row-major), rest 0
 *   [27] sum of squared correspondence distances, [28] number of correspondences,
 *   [29] number of pairs skipped for non-finite normals, [30..31] reserved */

typedef struct {
  int max_iterations;                     /* registration.h:566, default 10 */
  double max_correspondence_distance;     /* registration.h:117, default sqrt(DBL_MAX) */
  double transformation_epsilon;          /* registration.h:588, default 0 */
  double transformation_rotation_epsilon; /* registration.h:592, default 0 = unset */
  double euclidean_fitness_epsilon;       /* registration.h:116, default -DBL_MAX */
  int min_number_correspondences;         /* registration.h:621, default 3 */
  int mode;                               /* PCLHIP_ICP_POINT_TO_POINT / _PLANE */
  int failure_after_max_iterations;       /* default_convergence_criteria.h:294 */
  int max_iterations_similar_transforms;  /* default_convergence_criteria.h:318 */
  double mse_threshold_absolute;          /* default_convergence_criteria.h:310, 1e-12 */
} pclhip_icp_params;

typedef struct {
  float final_transformation[16];   /* row-major 4x4 */
  float last_transformation[16];    /* transformation_ of the last iteration */
  int nr_iterations;
  int converged;
  int convergence_state;            /* DefaultConvergenceCriteria::ConvergenceState values */
  uint64_t num_correspondences;     /* of the last iteration */
  double mse;                       /* mean squared correspondence distance, last iteration */
  double gpu_ms;                    /* GPU time of all iterate launches (events on ctx stream) */
  double gpu_ms_search_kernel;      /* ... of which the search + accumulate kernels of the iterations */
} pclhip_icp_result;

/* Optional hook run on the 32-double DEVICE record of every iteration before the host reads it:
 * multi-GPU runs all-reduce (sum) it over RCCL here.  `stream` is the context's hipStream_t on
 * which the record was produced.  Return 0 on success. */
typedef int (*pclhip_allreduce_fn)(void* user, double* device_sums, int count, void* stream);

PCLHIP_API void pclhip_icp_params_default(pclhip_icp_params* p);

/* Registration object bound to a target index.  Replaces the state held by
 * pcl::IterativeClosestPoint (registration/include/pcl/registration/icp.h:98-347). */
PCLHIP_API pclhip_status pclhip_icp_create(pclhip_index* target, pclhip_icp** out);
PCLHIP_API void pclhip_icp_destroy(pclhip_icp* icp);
/* Registration::setInputSource (registration.h:195-196): uploads + kd-orders the source. */
PCLHIP_API pclhip_status pclhip_icp_set_source(pclhip_icp* icp, const void* points,
                                               size_t stride_bytes, uint64_t n);
/* The same with PCLBase::setIndices (common/include/pcl/pcl_base.h:102-125) / CorrespondenceEstimationBase::
 * setIndicesSource (registration/include/pcl/registration/correspondence_estimation.h:194): only
 * points[indices[j]] take part in the correspondence search and the estimation; correspondences still carry
 * index_query into the ORIGINAL cloud.  indices == NULL: every point.  (The registered output of align() is
 * the whole input cloud moved by the final transformation either way, impl/icp.hpp:264-267.) */
PCLHIP_API pclhip_status pclhip_icp_set_source_indexed(pclhip_icp* icp, const void* points, size_t stride_bytes,
                                                       uint64_t n, const int32_t* indices, uint64_t n_indices);
/* Normals of the source cloud, one record per source point in the order given to
 * pclhip_icp_set_source (a pcl::PointNormal source: normals = points + 16, stride 48).  Required by
 * PCLHIP_ICP_SYMMETRIC; call after pclhip_icp_set_source. */
PCLHIP_API pclhip_status pclhip_icp_set_source_normals(pclhip_icp* icp, const void* normals, size_t stride_bytes);
/* setEnforceSameDirectionNormals (icp.h:416-428), default 1 */
PCLHIP_API pclhip_status pclhip_icp_set_enforce_same_direction_normals(pclhip_icp* icp, int enforce);
PCLHIP_API pclhip_status pclhip_icp_set_allreduce(pclhip_icp* icp, pclhip_allreduce_fn fn, void* user);
/* Rewind the working copy of the source to the input cloud (start of computeTransformation). */
PCLHIP_API pclhip_status pclhip_icp_reset(pclhip_icp* icp);

/* Correspondence rejectors applied on the device between the search and the accumulation of every
 * iteration (the chain of impl/icp.hpp:187-201), in the given order:
 *   DISTANCE         registration/src/correspondence_rejection_distance.cpp:43-68      param = max distance
 *   MEDIAN_DISTANCE  registration/src/correspondence_rejection_median_distance.cpp:43-69 param = factor
 *   ONE_TO_ONE       registration/src/correspondence_rejection_one_to_one.cpp:43-66
 *   TRIMMED          registration/src/correspondence_rejection_trimmed.cpp:43-60        param = overlap ratio
 * Exact ties, whose order the reference leaves to an unstable sort, go to the lower query index.
 * Multi-GPU (pclhip_icp_set_comm / _set_allreduce): DISTANCE is per pair; MEDIAN_DISTANCE and TRIMMED cut at the one
 * cloud-global order statistic (the histograms of their selection are all-reduced: the hook / communicator sees buffers
 * of 2048 doubles besides the 32-double record); ONE_TO_ONE resolves its conflicts over the ranks by a minimum
 * all-reduce of the per-target (distance, query) keys -- with the native communicator and a sharded TARGET
 * (pclhip_icp_set_region: every rank holds the whole source, so query indices are global; 8 bytes per target point
 * travel per iteration); with source slabs or an all-reduce hook it is refused.  PRECONDITION of that mode: the keys are
 * indexed by the ORIGINAL target index, so every rank must have built its index over the same WHOLE cloud plus its own
 * subset list (pclhip_index_build(points, ..., indices = pclhip_select_region(...)), as pcl_amd.dist.ShardedTarget does) --
 * not over a cloud that holds only its slab; the first iteration checks that all ranks report the same cloud size and
 * returns PCLHIP_ERR_STATE on every rank otherwise.  A rank whose share
 * of the source is empty still issues those collectives (with zero histograms), in step with its peers.  With the SOURCE
 * cut into slabs, TRIMMED breaks exact distance ties at the cut by the rank-LOCAL query index (every rank's indices
 * restart at 0): the number of pairs kept is the single-GPU run's, which of several exactly tied pairs survive may not
 * be. */
enum { PCLHIP_REJ_DISTANCE = 0, PCLHIP_REJ_MEDIAN_DISTANCE = 1, PCLHIP_REJ_ONE_TO_ONE = 2, PCLHIP_REJ_TRIMMED = 3 };
typedef struct {
  int kind;
  double param;
  uint32_t min_correspondences; /* TRIMMED: nr_min_correspondences_ */
  uint32_t reserved;
} pclhip_rejector;
PCLHIP_API pclhip_status pclhip_icp_set_rejectors(pclhip_icp* icp, const pclhip_rejector* list, int n);
/* median found by the last MEDIAN_DISTANCE rejector (getMedianDistance()) */
PCLHIP_API double pclhip_icp_last_median_distance(const pclhip_icp* icp);
/* use_reciprocal_correspondence_ (registration.h / impl/correspondence_estimation.hpp:220-311): keep
 * (i, m) only if the nearest source point of target[m] is i again (the reference rebuilds the source tree per iteration;
 * here the source index is built once, refitted to the moved cloud and searched from its root: same answers).
 * Multi-GPU: supported when the TARGET is sharded (pclhip_icp_set_region: every rank holds the whole source), refused
 * when the source is cut into slabs. */
PCLHIP_API pclhip_status pclhip_icp_set_reciprocal(pclhip_icp* icp, int enable);

/* One iteration on the device-resident working source cloud (a search kernel and a streaming
 * accumulation kernel):
 *   cur <- T_prev * cur   (transformCloud, impl/icp.hpp:49-111 / transforms.hpp:109-123)
 *   1-NN of every cur point in the target, drop d2 > max_dist^2
 *        (CorrespondenceEstimation::determineCorrespondences, impl/correspondence_estimation.hpp:145-218)
 *   accumulate the normal system (TransformationEstimationPointToPlaneLLS::estimateRigidTransformation,
 *        impl/transformation_estimation_point_to_plane_lls.hpp:165-245) or the umeyama sums.
 * T_prev: row-major 4x4 float (identity on the first iteration, or the guess).
 * sums: PCLHIP_ICP_NSUMS doubles on the host. */
PCLHIP_API pclhip_status pclhip_icp_iterate(pclhip_icp* icp, const float T_prev[16], double max_dist,
                                            int mode, double sums[PCLHIP_ICP_NSUMS]);

/* GPU time (ms, HIP events on the context stream) of the search + accumulate kernels of the
 * last pclhip_icp_iterate call. */
PCLHIP_API double pclhip_icp_last_kernel_ms(const pclhip_icp* icp);
/* GPU time (ms) the last pclhip_icp_set_source spent ordering the source spatially (the same ordering the
 * index build applies to the target; paid once per source cloud, not per iteration). */
PCLHIP_API double pclhip_icp_source_order_ms(const pclhip_icp* icp);
/* Duration (ms) of the search kernel alone in the last pclhip_icp_iterate (the default iteration is
 * two kernels: search, then the streaming accumulation of the 6x6 / umeyama sums). */
PCLHIP_API double pclhip_icp_last_search_ms(const pclhip_icp* icp);

/* Host-side closed forms on a reduction record (exposed for the adapters and for tests):
 * 6x6 solve + constructTransformationMatrix (…point_to_plane_lls.hpp:132-163,264-268) or umeyama
 * (common/include/pcl/common/impl/eigen.hpp:675-738). */
PCLHIP_API pclhip_status pclhip_solve_transformation(const double sums[PCLHIP_ICP_NSUMS], int mode,
                                                     float T[16]);

/* The whole loop of IterativeClosestPoint::computeTransformation (impl/icp.hpp:113-268) with
 * DefaultConvergenceCriteria (impl/default_convergence_criteria.hpp:49-140).  The criteria's
 * previous-MSE memory persists across calls on the same object, as in the reference.
 * guess: row-major 4x4 or NULL. */
PCLHIP_API pclhip_status pclhip_icp_align(pclhip_icp* icp, const pclhip_icp_params* params,
                                          const float guess[16], pclhip_icp_result* result);

/* DefaultConvergenceCriteria::hasConverged (impl/default_convergence_criteria.hpp:49-140) as a host function, for
 * callers that run the loop themselves (a Registration with a foreign CorrespondenceEstimation /
 * TransformationEstimation plugged in): the same code the device loop runs.  `params` supplies the thresholds as
 * ICP sets them (impl/icp.hpp:157-161); `st` is the criteria's memory, which persists across alignments.
 * Returns 1 when the alignment ends with converged_ = true; the loop also ends whenever
 * st->convergence_state != 0 (NOT_CONVERGED). */
typedef struct {
  double prev_mse;                    /* correspondences_prev_mse_, DBL_MAX initially */
  int iterations_similar_transforms;
  int convergence_state;              /* ConvergenceState values of default_convergence_criteria.h:71-80 */
} pclhip_convergence_state;
PCLHIP_API void pclhip_convergence_init(pclhip_convergence_state* st);
PCLHIP_API int pclhip_convergence_has_converged(const pclhip_icp_params* params, pclhip_convergence_state* st,
                                                int nr_iterations, const float T[16], double mse);

/* The same loop as a stream of exactly n_steps iterations for measurements and for back-to-back registration
 * of the same clouds: Registration::align() called again and again -- when an alignment ends (converged or
 * not) the next step starts the next one from the input cloud and `guess`, on the device, without host
 * involvement.  All steps are queued on the context's stream before the first result is read; the solve,
 * final = T * final and the convergence test of every iteration run in a device kernel.  out_steps[i]
 * describes step i.  No rejectors / reciprocal correspondences (they need host decisions). */
typedef struct {
  int iteration;              /* nr_iterations_ of the running alignment after this step (0: no correspondences) */
  int convergence_state;      /* DefaultConvergenceCriteria state after this step */
  int converged;              /* the alignment ended here with converged_ = true */
  int alignment_ended;        /* this step was the last of its alignment */
  uint64_t num_correspondences;
  double mse;
  float search_ms;            /* HIP events on the context's stream: the search kernel ... */
  float kernels_ms;           /* ... search + accumulation ... */
  float step_ms;              /* ... the whole step incl. reduction, (all-reduce,) solve */
  float final_transformation[16];
} pclhip_icp_step;
PCLHIP_API pclhip_status pclhip_icp_run_steps(pclhip_icp* icp, const pclhip_icp_params* params,
                                              const float guess[16], int n_steps, pclhip_icp_step* out_steps);

/* ---- multi-GPU: one process per GPU, the 32-double record summed over the ranks per iteration --------
 * PCL has no multi-GPU path (gpu/containers/src/initialization.cpp:109 picks one device); the contract is
 * SURVEY.md 8(e).  A communicator wraps an RCCL (ncclComm_t) group over xGMI: rank 0 obtains an id, the
 * application distributes its 128 bytes to the other ranks by whatever means it has (MPI, a file, a socket,
 * torch.distributed's store), every rank calls pclhip_comm_create.  With a communicator attached the
 * all-reduce is issued from C on the context's stream between the reduction and the solve kernel of every
 * iteration; RCCL is bound with dlopen at first use (no link-time dependency). */
#define PCLHIP_COMM_ID_BYTES 128
typedef struct pclhip_comm pclhip_comm;
PCLHIP_API pclhip_status pclhip_comm_get_unique_id(unsigned char id[PCLHIP_COMM_ID_BYTES]);
PCLHIP_API pclhip_status pclhip_comm_create(pclhip_ctx* ctx, int rank, int nranks,
                                            const unsigned char id[PCLHIP_COMM_ID_BYTES], pclhip_comm** out);
PCLHIP_API void pclhip_comm_destroy(pclhip_comm* comm);
PCLHIP_API int pclhip_comm_rank(const pclhip_comm* comm);
PCLHIP_API int pclhip_comm_size(const pclhip_comm* comm);
/* in-place sum of `count` doubles in device memory over the ranks, on the context's stream (asynchronous) */
PCLHIP_API pclhip_status pclhip_comm_allreduce_sum_f64(pclhip_comm* comm, double* device_buf, int count);
/* attach (or detach with NULL) a communicator; replaces the pclhip_icp_set_allreduce hook when both are set */
PCLHIP_API pclhip_status pclhip_icp_set_comm(pclhip_icp* icp, pclhip_comm* comm);

/* ---- target sharding: a target cloud spread over the GPUs of a node ------------------------
 * The target is cut into n_slabs cells of a kd partition of space (recursive bisection at order statistics along
 * the widest axis: equal point counts, half-open boxes [lo, hi) that tile R^3, unbounded on the outside).  Rank g
 *   1. selects the target points of its region dilated by a margin >= max_correspondence_distance (the halo) and
 *      builds its index on them -- pclhip_index_build with that index list, so results keep ORIGINAL indices;
 *   2. holds the whole source, but serves only the points whose CURRENT position lies in its region
 *      (pclhip_icp_set_region): such a point finds its true nearest neighbour in the rank's index whenever that
 *      neighbour is within the margin, and correctly finds none within max_correspondence_distance otherwise;
 *   3. takes part in the per-iteration all-reduce of the record (pclhip_icp_set_comm); all ranks then apply the
 *      same transformation, so routing stays consistent without any other exchange.
 * Correspondences over all ranks are exactly those of a single index over the whole target.
 * Partitioning and selection are host code (no GPU needed); clouds may be host or device memory.  A cloud in DEVICE
 * memory is cut on the device (same regions and lists, bit for bit; up to 254 slabs, more fall back to the host code):
 * these two calls take no context, run on the null stream and first wait for everything queued on the device, so a
 * cloud another stream is still producing is complete when they read it.
 * regions: n_slabs x 6 floats (lo.xyz, hi.xyz), +-inf on unbounded sides. */
PCLHIP_API pclhip_status pclhip_partition_slabs(const void* points, size_t stride_bytes, uint64_t n, int n_slabs,
                                                float* regions);
/* ascending indices of the finite points inside `region` dilated by `margin` per axis (rounded outwards);
 * *out_count is always set; PCLHIP_ERR_OVERFLOW if it exceeds `capacity` */
PCLHIP_API pclhip_status pclhip_select_region(const void* points, size_t stride_bytes, uint64_t n,
                                              const float region[6], double margin, int32_t* out_indices,
                                              uint64_t capacity, uint64_t* out_count);
/* the slab whose region holds the point (x >= lo && x < hi per axis), -1 for non-finite points */
PCLHIP_API int pclhip_region_owner(const float* regions, int n_slabs, const float xyz[3]);
/* restrict the registration to the source points whose current position lies in `region`; NULL: all points.  Inside
 * pclhip_icp_align / pclhip_icp_run_steps the rank can WALK only the 64-point groups of its (kd-ordered) source copy
 * whose box touches the region -- about n / n_slabs points per iteration, not n; groups that come into reach later are
 * brought up to date from the transforms they missed, bit for bit (measured in round 4: one rank of eight of a 30M-point
 * job iterates in 0.59 ms against 1.82 ms for the full pass).  pclhip_ctx_set_option(ctx, "served_groups", 0) walks the
 * whole source copy and masks per point -- the reference the tests compare with */
PCLHIP_API pclhip_status pclhip_icp_set_region(pclhip_icp* icp, const float region[6]);
/* Largest squared distance to the k-th nearest neighbour (the point itself counts as the first, as in
 * pclhip_normals) over the indexed points inside `box` (lo.xyz, hi.xyz; NULL: all).  With a halo index this
 * tells whether the normals of the points a rank can be matched to are exact: they are when
 * sqrt(*out_max_d2) <= margin - max_correspondence_distance (every neighbour then lies inside the halo). */
PCLHIP_API pclhip_status pclhip_index_kth_distance_max(pclhip_index* index, int k, const float box[6],
                                                       double* out_max_d2);

/* Registration::getFitnessScore(max_range) (registration/include/pcl/registration/impl/registration.hpp:132-168):
 * the source is transformed by T (row-major 4x4, Transformer::se3 operation order like
 * transformPointCloud), every finite point looks up its nearest target point, and the SQUARED
 * distances that are <= max_range (compared as given, like the reference) are averaged in double.
 * *score = DBL_MAX when nothing qualifies; *nr (may be NULL) = number of points that counted.
 * Does not disturb the state of the alignment (seeds, correspondences of the last iteration). */
PCLHIP_API pclhip_status pclhip_icp_fitness_score(pclhip_icp* icp, const float T[16], double max_range,
                                                  double* score, uint64_t* nr);

/* Correspondences of the LAST iteration, materialised lazily as pcl::Correspondences
 * (common/include/pcl/correspondence.h:60-89): sorted by index_query, entries beyond max_dist
 * omitted.  With rejectors the list is the chain's output in the reference's order (ONE_TO_ONE:
 * by match index, TRIMMED: by distance).  Buffers (host or device) must hold n_source entries;
 * *out_n receives the count. */
PCLHIP_API pclhip_status pclhip_icp_fetch_correspondences(pclhip_icp* icp, int32_t* index_query,
                                                          int32_t* index_match, float* distance,
                                                          uint64_t* out_n);
/* The same list as an array of pcl::Correspondence records -- { int32 index_query; int32 index_match; float distance },
 * 12 bytes (correspondence.h:60-71) --, compacted on the device and handed over in ONE copy: what
 * CorrespondenceEstimation::determineCorrespondences leaves in its std::vector (impl/correspondence_estimation.hpp
 * :160-216: resized to the number of queries, filled, shrunk to the valid ones).  Query order only: refused (ERR_STATE)
 * when a ONE_TO_ONE or TRIMMED rejector re-orders the list.  `capacity` records at `out` (host or device); *out_n receives
 * the count, PCLHIP_ERR_OVERFLOW if it exceeds the capacity.  10M pairs through the binding
 * (CorrespondenceEstimationHIP::determineCorrespondences on host clouds): 102 -> 13.6 ms repeated, 138 -> 66 ms the first
 * time (scratch/boundary_probe.cpp). */
PCLHIP_API pclhip_status pclhip_icp_fetch_correspondence_records(pclhip_icp* icp, void* out, uint64_t capacity,
                                                                 uint64_t* out_n);

/* TransformationEstimation::estimateRigidTransformation(cloud_src, cloud_tgt, T)
 * (registration/include/pcl/registration/transformation_estimation.h:71-115) for n explicit pairs
 * (src[i], tgt[i]); `mode` selects TransformationEstimationSVD / PointToPlaneLLS (target normals) /
 * SymmetricPointToPlaneLLS (both normals).  Buffers host or device; normals may be NULL when the mode
 * does not use them.  T: row-major 4x4; sums (may be NULL): the PCLHIP_ICP_NSUMS reduction record. */
PCLHIP_API pclhip_status pclhip_estimate_rigid_transformation(
    pclhip_ctx* ctx, int mode, const void* src, size_t src_stride, const void* src_normals,
    size_t src_normals_stride, const void* tgt, size_t tgt_stride, const void* tgt_normals,
    size_t tgt_normals_stride, uint64_t n, int enforce_same_direction_normals, float T[16], double* sums);

/* TransformationEstimationPointToPlaneLLSWeighted::estimateRigidTransformation
 * (impl/transformation_estimation_point_to_plane_lls_weighted.hpp:195-290): the point-to-plane system
 * with every target normal scaled by its pair's weight (n floats, host or device). */
PCLHIP_API pclhip_status pclhip_estimate_rigid_transformation_weighted(
    pclhip_ctx* ctx, const void* src, size_t src_stride, const void* tgt, size_t tgt_stride,
    const void* tgt_normals, size_t tgt_normals_stride, const float* weights, uint64_t n, float T[16],
    double* sums);

/* out = T * in for n records (x,y,z at byte 0; other bytes of the record untouched);
 * order 0: Eigen Matrix4f*Vector4f order (icp.hpp:49-111); order 1: Transformer::se3 order
 * (transforms.hpp:117-123).  If normals_offset_bytes != 0 the 3 floats there are rotated too. */
PCLHIP_API pclhip_status pclhip_transform_cloud(pclhip_ctx* ctx, const float T[16], int order,
                                                const void* in, void* out, size_t stride_bytes,
                                                uint64_t n, size_t normals_offset_bytes);
/* The same for the SOURCE cloud of a registration: when `in` is the host buffer pclhip_icp_set_source[_indexed] was
 * given (same address, stride and count), the records staged on the device then are used and nothing is uploaded
 * again -- the moved cloud IterativeClosestPoint::computeTransformation hands back (impl/icp.hpp:264-267:
 * output = *input_, then transformCloud) costs one download.  Any other `in` behaves like pclhip_transform_cloud.
 * CONTRACT: the staged records are the cloud as it was when pclhip_icp_set_source[_indexed] was called -- like the
 * registration's own copy of the points.  A caller that edits the buffer in place (or reuses the address for another
 * cloud) must call pclhip_icp_set_source[_indexed] again before aligning or transforming, exactly as it must for the
 * alignment itself to see the change; the copy is dropped there and is never kept beyond 1/8 of the device's memory. */
PCLHIP_API pclhip_status pclhip_icp_transform_source(pclhip_icp* icp, const float T[16], int order,
                                                     const void* in, void* out, size_t stride_bytes,
                                                     uint64_t n, size_t normals_offset_bytes);

/* ---- VoxelGrid ----------------------------------------------------------------------------------
 * Replaces pcl::VoxelGrid<pcl::PointXYZ>::applyFilter (filters/include/pcl/filters/impl/
 * voxel_grid.hpp:597-814) with downsample_all_data and the optional pass-through filter in front of
 * the grid (setFilterFieldName + setFilterLimits + setFilterLimitsNegative, voxel_grid.h:440-476;
 * impl/voxel_grid.hpp:513-590,684-695).  `has_z_limits`: 0 = no filter; 1 = keep z inside [z_min, z_max];
 * generally bit 0 = on, bit 1 = negative (cut the points INSIDE the interval instead), bits 8..15 = 1 + position
 * of the filter field inside the record, counted in floats (0 = the z coordinate; PCLHIP_VOXELGRID_LIMITS
 * composes it).  Output: ascending voxel id, records x,y,z,1
 * (16 B).  out must hold n records; *out_n receives the count.  Returns PCLHIP_ERR_OVERFLOW when
 * the reference would refuse (:620-629). */
#define PCLHIP_VOXELGRID_LIMITS(float_index_of_field, negative) (1 | ((negative) ? 2 : 0) | (((float_index_of_field) + 1) << 8))
PCLHIP_API pclhip_status pclhip_voxelgrid(pclhip_ctx* ctx, const void* points, size_t stride_bytes,
                                          uint64_t n, const float leaf[3], uint32_t min_points_per_voxel,
                                          int has_z_limits, double z_min, double z_max,
                                          void* out_xyzw, uint64_t* out_n);

/* The same for records that carry normals and for setDownsampleAllData (filters/include/pcl/filters/voxel_grid.h
 * :293-302, impl/voxel_grid.hpp:790-809).  Input and output records share a layout: x y z at +0 and, when
 * normals_offset != 0 (pcl::PointNormal: 16, strides 48), normal[3], 0, curvature, 0 0 0 at +normals_offset.
 *   downsample_all_data != 0: the CentroidPoint accumulators of common/include/pcl/common/impl/accumulators.hpp
 *        :68-127 -- coordinates averaged, normal = normalised sum of the voxel's normals, curvature averaged;
 *   downsample_all_data == 0: only the coordinates are averaged, every other field of the output keeps its
 *        default (0).
 * Sums run in ascending input index inside a voxel.  out must hold n records of out_stride bytes. */
PCLHIP_API pclhip_status pclhip_voxelgrid_ex(pclhip_ctx* ctx, const void* points, size_t stride_bytes, uint64_t n,
                                             const float leaf[3], uint32_t min_points_per_voxel, int has_z_limits,
                                             double z_min, double z_max, int downsample_all_data,
                                             size_t normals_offset, void* out, size_t out_stride_bytes,
                                             uint64_t* out_n);

/* The grid of a filter run: getMinBoxCoordinates / getMaxBoxCoordinates / getNrDivisions / getDivisionMultiplier
 * (filters/include/pcl/filters/voxel_grid.h:326-344).  Cell (i, j, k) has the id
 * (i - min_b[0]) * divb_mul[0] + (j - min_b[1]) * divb_mul[1] + (k - min_b[2]) * divb_mul[2]. */
typedef struct pclhip_voxelgrid_dims {
  int32_t min_b[3];
  int32_t max_b[3];
  int32_t div_b[3];
  int32_t divb_mul[3];
} pclhip_voxelgrid_dims;

/* pclhip_voxelgrid_ex plus setSaveLeafLayout (voxel_grid.h:316, impl/voxel_grid.hpp:752-787): leaf_layout (host
 * or device, may be NULL) receives, for every cell id, the position of that voxel's centroid in the output or -1
 * for a cell that is empty or was dropped by min_points_per_voxel; it must hold div_b[0]*div_b[1]*div_b[2] ints
 * (leaf_layout_capacity is checked).  dims (may be NULL) receives the grid.  Call pclhip_voxelgrid_grid first to
 * learn how many cells the layout has. */
PCLHIP_API pclhip_status pclhip_voxelgrid_ex2(pclhip_ctx* ctx, const void* points, size_t stride_bytes, uint64_t n,
                                              const float leaf[3], uint32_t min_points_per_voxel, int has_z_limits,
                                              double z_min, double z_max, int downsample_all_data,
                                              size_t normals_offset, void* out, size_t out_stride_bytes,
                                              uint64_t* out_n, int32_t* leaf_layout, uint64_t leaf_layout_capacity,
                                              pclhip_voxelgrid_dims* dims);

/* Only the grid the filter would use for this cloud (bounding box pass, impl/voxel_grid.hpp:613-645). */
PCLHIP_API pclhip_status pclhip_voxelgrid_grid(pclhip_ctx* ctx, const void* points, size_t stride_bytes, uint64_t n,
                                               const float leaf[3], int has_z_limits, double z_min, double z_max,
                                               pclhip_voxelgrid_dims* dims);

/* ---- PCD files (io/src/pcd_io.cpp: PCDReader :115-675, PCDWriter :848-1500) ----------------------
 * The on-disk format either side of the path.  ascii, binary and binary_compressed (LZF, fields stored
 * as struct of arrays) are read straight into the strided records the calls above consume.  Errors are
 * reported through pclhip_last_error(NULL). */
typedef struct {
  uint64_t points;       /* POINTS */
  uint32_t width, height;
  int data_type;         /* 0 ascii, 1 binary, 2 binary_compressed */
  int version;           /* 7 when a VIEWPOINT line is present, else 6 */
  uint32_t point_step;   /* bytes per point in the file */
  uint32_t num_fields;
  int has_xyz, has_normals, has_curvature, has_intensity, has_rgb;
  float viewpoint[7];    /* tx ty tz qw qx qy qz */
  uint64_t data_offset;  /* byte offset of the body */
} pclhip_pcd_info;

/* PCDReader::readHeader */
PCLHIP_API pclhip_status pclhip_pcd_read_header(const char* path, pclhip_pcd_info* info);
/* pcl::io::loadPCDFile into records of stride_bytes: x,y,z at +0 (and 1.0f at +12 when the record has
 * room, like PointXYZ); if normals_offset != 0 and the file has normal_x/y/z they go to +normals_offset
 * (3 floats, then 0.0f, then curvature at +normals_offset+16 when the file has it and the record has
 * room -- the PointNormal layout with normals_offset = 16, stride 48).  float64 / integer fields are
 * converted.  points: host or device memory holding `capacity` records; *n_out = POINTS (also when the
 * buffer is too small: PCLHIP_ERR_OVERFLOW); *is_dense as the reference computes it. */
PCLHIP_API pclhip_status pclhip_pcd_read(const char* path, void* points, size_t stride_bytes,
                                         size_t normals_offset, uint64_t capacity, uint64_t* n_out,
                                         int* is_dense);
/* pcl::io::savePCDFile{ASCII,Binary,BinaryCompressed}: unorganized cloud of n records (x y z, plus
 * normal_x normal_y normal_z [curvature] when normals_offset != 0); precision: significant digits of the
 * ascii writer (<= 0: the reference's default 8). */
PCLHIP_API pclhip_status pclhip_pcd_write(const char* path, const void* points, size_t stride_bytes,
                                          size_t normals_offset, uint64_t n, int data_type, int precision);
/* The same for an organized cloud (width x height records, row-major) with an acquisition pose
 * (VIEWPOINT tx ty tz qw qx qy qz; NULL = 0 0 0 1 0 0 0). */
PCLHIP_API pclhip_status pclhip_pcd_write_organized(const char* path, const void* points, size_t stride_bytes,
                                                    size_t normals_offset, uint32_t width, uint32_t height,
                                                    const float viewpoint[7], int data_type, int precision);
/* One component of any field of the file (e.g. "intensity", "curvature", "normal_x") as float32, one value
 * per point, into HOST memory.  float64 / integer fields are converted; 4-byte "rgb" / "rgba" keep their
 * 32 bits (PCL packs colours into a float). */
PCLHIP_API pclhip_status pclhip_pcd_read_field(const char* path, const char* field, uint32_t component,
                                               float* out, uint64_t capacity, uint64_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* PCLHIP_H_ */
