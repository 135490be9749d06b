/*
 * pcl_oracle.h -- CPU restatement of the PCL ICP hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the *checker* for the MI355X implementation in pcl_amd/csrc.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product path never does.
 *
 * Parity status: PINNED.  The reference itself cannot be compiled in this container (needs Eigen,
 * Boost, FLANN -- none installed, no network), so there is no oracle/_ref build.  The restatement
 * is pinned against the reference's own golden vectors (tests/golden/, extracted by
 * tests/golden/make_golden.py from /root/reference/test): 397/397 + 53/53 bunny correspondences,
 * the ICP 4x4 golden (1e-3), k=10 hand-point orders, the bun0 plane-fit normal (1e-4), the
 * VoxelGrid 103/14 counts, the LLS and SVD known-answer tests.
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 * Points are passed as float arrays with a stride in FLOATS: p[i*stride + 0..2] = x,y,z.
 * All floating-point evaluation orders are fixed and compiled with -ffp-contract=off.
 */
#ifndef PCL_ORACLE_H_
#define PCL_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- exact k-NN --------------------------------------------------------------------------- */
/* L2_Simple float distance ((dx*dx)+dy*dy)+dz*dz (FLANN 1.9.1 dist.h L2_Simple, call sites
 * kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp:154-203).  Results ascending by (d2, index);
 * ties keep the LOWER index (search/include/pcl/search/impl/brute_force.hpp:92,115 strict '>').
 * k is clamped to the number of finite target points (kdtree_flann.hpp:241-242); unused output
 * slots are filled with index -1 / distance +inf.  Non-finite target points are dropped
 * (kdtree_flann.hpp:443-452); non-finite queries return 0 neighbours.  Returns k_effective. */
int orc_knn_bruteforce(const float* tgt, int64_t nt, int ts, const float* qry, int64_t nq, int qs,
                       int k, int32_t* out_idx, float* out_d2, int nthreads);

/* Exact kd-tree (leaf <= 15 points, split widest bbox dimension at the sliding midpoint -- the
 * published FLANN KDTreeSingleIndex scheme, params at kdtree_flann.hpp:131-135).  Used for
 * problem sizes brute force cannot finish and for CPU-baseline timing.  Identical results to
 * orc_knn_bruteforce by construction (lexicographic (d2,index) result set, strict pruning). */
typedef struct orc_kdtree orc_kdtree;
orc_kdtree* orc_kdtree_build(const float* pts, int64_t n, int stride);
void orc_kdtree_free(orc_kdtree* t);
int64_t orc_kdtree_size(const orc_kdtree* t);
int orc_kdtree_knn(const orc_kdtree* t, const float* qry, int64_t nq, int qs, int k,
                   int32_t* out_idx, float* out_d2, int nthreads);

/* ---- CorrespondenceEstimation::determineCorrespondences ---------------------------------- */
/* registration/include/pcl/registration/impl/correspondence_estimation.hpp:145-218.
 * For each source point (ascending index): skip non-finite, 1-NN, drop if d2 > max_dist^2
 * (double compare, :161,176).  Outputs sorted by query index.  Returns the number emitted. */
int64_t orc_correspondences(const orc_kdtree* t, const float* src, int64_t ns, int ss,
                            double max_dist, int32_t* out_q, int32_t* out_m, float* out_d2,
                            int nthreads);
/* Registration::getFitnessScore (registration/include/pcl/registration/impl/registration.hpp:132-168) */
double orc_fitness_score(const orc_kdtree* t, const float* src, int64_t ns, int ss, const float* T,
                         double max_range, int64_t* out_nr, int nthreads);
/* determineReciprocalCorrespondences (:220-311): keep (i, m) only if the 1-NN of target[m] in the
 * source tree is i again.  src_tree indexes the source cloud. */
int64_t orc_reciprocal_correspondences(const orc_kdtree* tgt_tree, const orc_kdtree* src_tree,
                                       const float* src, int64_t ns, int ss, const float* tgt,
                                       int ts, double max_dist, int32_t* out_q, int32_t* out_m,
                                       float* out_d2, int nthreads);

/* ---- transformation estimation ------------------------------------------------------------ */
/* TransformationEstimationPointToPlaneLLS (impl/transformation_estimation_point_to_plane_lls.hpp
 * :165-268): float products a,b,c,d, double accumulation of 21+6 terms in correspondence order,
 * mirror, x = ATA^-1 * ATb (partial-pivot LU inverse, as Eigen's 6x6 inverse()), then
 * constructTransformationMatrix (:132-163).  nrm = target normals (stride ns_ floats).
 * sums27 (optional) receives the 21 upper-triangular ATA terms (row-major order of :213-233)
 * followed by the 6 ATb terms.  T is row-major 4x4 float.  Returns pairs used. */
int64_t orc_lls_point_to_plane(const float* src, int ss, const float* tgt, int ts,
                               const float* nrm, int ns_, const int32_t* q, const int32_t* m,
                               int64_t npairs, double* sums27, float* T);
void orc_lls_solve(const double* sums27, float* T);

/* TransformationEstimationSymmetricPointToPlaneLLS
 * (impl/transformation_estimation_symmetric_point_to_plane_lls.hpp:149-197); see the .c file. */
int64_t orc_lls_symmetric(const float* src, int ss, const float* src_nrm, int sns, const float* tgt,
                          int ts, const float* tgt_nrm, int tns, const int32_t* q, const int32_t* m,
                          int64_t npairs, int enforce_same_direction, int acc_double, double* sums27,
                          float* T);
void orc_symmetric_solve(const double* sums27, float* T);

/* TransformationEstimationSVD with use_umeyama_ = true (impl/transformation_estimation_svd.hpp
 * :127-155 -> common/include/pcl/common/impl/eigen.hpp:675-738).  acc_double = 0: all sums in
 * float, sequential (Scalar = float as in the reference; Eigen's internal summation order is not
 * reproducible, sequential is the in-tree restated order); acc_double = 1: sums in double (the
 * exact-arithmetic limit, used when comparing at >= 1M points where float-sum noise ~1e-5). */
int64_t orc_umeyama(const float* src, int ss, const float* tgt, int ts, const int32_t* q,
                    const int32_t* m, int64_t npairs, int acc_double, float* T);
/* Same closed form from the 15 raw sums (sum s[3], sum t[3], sum t_i*s_j [9] row-major) + count. */
void orc_umeyama_from_sums(const double* sums15, double count, float* T);

/* ---- transformCloud ----------------------------------------------------------------------- */
/* order 0: IterativeClosestPoint::transformCloud (impl/icp.hpp:49-111): Eigen Matrix4f*Vector4f,
 *          r = ((c0*x + c1*y) + c2*z) + c3*1  per row, no FMA.
 * order 1: IterativeClosestPointWithNormals::transformCloud -> transformPointCloudWithNormals ->
 *          Transformer<float>::se3 (common/include/pcl/common/impl/transforms.hpp:117-123):
 *          r = c0*x + (c1*y + (c2*z + c3)); normals via so3 (:109-115): c0*x + (c1*y + c2*z).
 * Non-finite points are left untouched.  In-place allowed.  nrm_in/out may be NULL. */
void orc_transform_cloud(const float* T, int order, const float* in, int is_, float* out, int os,
                         const float* nrm_in, int nis, float* nrm_out, int nos, int64_t n);
/* final = T * final in float (impl/icp.hpp:223), coefficient order ((a0b0+a1b1)+a2b2)+a3b3. */
void orc_mat4_mul(const float* A, const float* B, float* C);

/* ---- DefaultConvergenceCriteria ----------------------------------------------------------- */
/* impl/default_convergence_criteria.hpp:49-140 + default_convergence_criteria.h:283-318.
 * State values match the reference enum (default_convergence_criteria.h:71-80). */
enum {
  ORC_NOT_CONVERGED = 0, ORC_ITERATIONS = 1, ORC_TRANSFORM = 2, ORC_ABS_MSE = 3, ORC_REL_MSE = 4,
  ORC_NO_CORRESPONDENCES = 5, ORC_FAILURE_AFTER_MAX_ITERATIONS = 6
};
typedef struct {
  int max_iterations;                      /* 1000 by default in the criteria; ICP sets it */
  int failure_after_max_iter;              /* false */
  double rotation_threshold;               /* 0.99999 */
  double translation_threshold;            /* 3e-4*3e-4 */
  double mse_threshold_relative;           /* 1e-5 */
  double mse_threshold_absolute;           /* 1e-12 */
  int max_iterations_similar_transforms;   /* 0 */
  /* state */
  int iterations_similar_transforms;
  double correspondences_prev_mse;         /* DBL_MAX */
  double correspondences_cur_mse;          /* DBL_MAX */
  int convergence_state;
} orc_convergence;
void orc_convergence_init(orc_convergence* c);
/* iterations = nr_iterations_ (already incremented), T = this iteration's transformation_,
 * mse = mean of the correspondence distances (calculateMSE, default_convergence_criteria.h:262-270). */
int orc_convergence_has_converged(orc_convergence* c, int iterations, const float* T, double mse);

/* ---- the ICP loop -------------------------------------------------------------------------- */
typedef struct {
  int max_iterations;                   /* registration.h:566 default 10 */
  double max_correspondence_distance;   /* registration.h:117 sqrt(DBL_MAX) */
  double transformation_epsilon;        /* 0 */
  double transformation_rotation_epsilon; /* 0 (unset) */
  double euclidean_fitness_epsilon;     /* -DBL_MAX */
  int min_number_correspondences;       /* 3 */
  int mode;                             /* 0 = point-to-point SVD (ICP), 1 = point-to-plane LLS (ICPWithNormals) */
  int acc_double;                       /* see orc_umeyama */
  int nthreads;
  int use_reciprocal;                   /* 0 */
} orc_icp_params;
typedef struct {
  float final_transformation[16];
  int nr_iterations;
  int converged;
  int convergence_state;
  int64_t last_num_correspondences;
  double last_mse;
  double seconds_search;                /* wall time in correspondence search */
  double seconds_total;
} orc_icp_result;
void orc_icp_params_default(orc_icp_params* p);
/* IterativeClosestPoint::computeTransformation (impl/icp.hpp:113-268).  tgt_nrm is required for
 * mode 1.  guess = row-major 4x4 (NULL = identity).  The target tree is built by the caller.
 * per_iter_T (optional, capacity max_iterations*16) records each iteration's transformation_.
 * per_iter_match (optional, capacity max_iterations*ns) records each iteration's 1-NN index per
 * source point (-1 when dropped) so per-iteration index parity can be checked. */
int orc_icp_align(const orc_kdtree* tgt_tree, const float* tgt, int ts, const float* tgt_nrm,
                  int tns, const float* src, int64_t ns, int ss, const float* guess,
                  const orc_icp_params* p, orc_convergence* conv, orc_icp_result* r,
                  float* per_iter_T, int32_t* per_iter_match);

/* GeneralizedIterativeClosestPoint::computeCovariances (impl/gicp.hpp:70-147); out: 9 doubles per point.
 * Returns -1 when the cloud has fewer than k points. */
int orc_gicp_covariances(const orc_kdtree* t, const float* cloud, int64_t n, int cs, int k, double eps,
                         double* out, int nthreads);

/* ---- NormalEstimation ---------------------------------------------------------------------- */
/* computeMeanAndCovarianceMatrix (common/include/pcl/common/impl/centroid.hpp:581-650), float. */
unsigned orc_mean_and_covariance(const float* cloud, int cs, const int32_t* indices, int n,
                                 float* cov9, float* centroid4);
/* solvePlaneParameters (features/include/pcl/features/impl/feature.hpp:64-89) + pcl::eigen33
 * (common/include/pcl/common/impl/eigen.hpp:295-325, computeRoots :68-128). */
void orc_solve_plane_parameters(const float* cov9, float* nx, float* ny, float* nz,
                                float* curvature);
/* NormalEstimation::computeFeature with k-NN (features/include/pcl/features/impl/normal_3d.hpp
 * :48-95) over the whole cloud with surface == input; flipNormalTowardsViewpoint
 * (normal_3d.h:169-188).  out = n x 4 floats (nx, ny, nz, curvature).  out_knn (optional) n*k
 * neighbour indices.  Returns the number of NaN normals. */
int64_t orc_normals_knn(const orc_kdtree* t, const float* cloud, int64_t n, int cs, int k,
                        const float* viewpoint3, float* out, int32_t* out_knn, int nthreads);

/* The same for the points listed in `indices` only (Feature::setIndices; output row j = cloud[indices[j]]);
 * the neighbours still come from the whole cloud (the search surface). */
int64_t orc_normals_knn_indices(const orc_kdtree* t, const float* cloud, int64_t n, int cs, int k,
                                const float* viewpoint3, const int32_t* indices, int64_t n_indices,
                                float* out, int32_t* out_knn, int nthreads);

/* ---- VoxelGrid ----------------------------------------------------------------------------- */
/* VoxelGrid::applyFilter (filters/include/pcl/filters/impl/voxel_grid.hpp:597-814), PointXYZ,
 * downsample_all_data (centroid = float sum / n, common/include/pcl/common/impl/accumulators.hpp
 * :68-85), optional z-field limits (has_limits: keep lim_min <= z <= lim_max).  Within a voxel the
 * summation order is ascending input index (stable sort; the reference's spreadsort order is
 * unspecified).  out must hold n*4 floats (x,y,z,1).  Returns the number of output points, or
 * -1 if the voxel grid would overflow int32 (the reference then returns the input unchanged). */
int64_t orc_voxelgrid(const float* cloud, int64_t n, int cs, const float* leaf3,
                      unsigned min_points_per_voxel, int has_limits, double lim_min, double lim_max,
                      float* out, int32_t* out_voxel_ids);

#ifdef __cplusplus
}
#endif
#endif
