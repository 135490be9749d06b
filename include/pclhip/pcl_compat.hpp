// pcl_compat.hpp -- header-only C++ mirror of the PCL plugin surface for the ICP hot path, written
// against the C ABI of pclhip.h (no Eigen, no Boost, no FLANN).  Names, argument meaning and error
// behaviour follow the reference classes so user code (and PCL's own tests) port by changing the
// namespace:
//
//   pclhip::PointXYZ / PointNormal / Normal      common/include/pcl/impl/point_types.hpp:315-321,843-853,787-794
//   pclhip::PointCloud<PointT>                   common/include/pcl/point_cloud.h:173,393-409
//   pclhip::Correspondence(s)                    common/include/pcl/correspondence.h:60-91
//   pclhip::search::KdTree<PointT>               search/include/pcl/search/kdtree.h:61-168
//   pclhip::registration::CorrespondenceEstimation  registration/include/pcl/registration/correspondence_estimation.h
//   pclhip::IterativeClosestPoint(+WithNormals)  registration/include/pcl/registration/icp.h:98-347,360-440
//   pclhip::NormalEstimation                     features/include/pcl/features/normal_3d.h:243-420
//   pclhip::VoxelGrid                            filters/include/pcl/filters/voxel_grid.h:221-533
//
// With real PCL available, the same calls sit inside subclasses of the real pcl:: bases; that binding
// is shown in INTEGRATION.md (it cannot be compiled in this image: PCL needs Eigen/Boost/FLANN).
// Errors follow PCL's convention: no exceptions on the hot path, `false`/0 results + a message
// (getLastError()), see SURVEY.md 8(b).
#pragma once

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <limits>
#include <vector>

#include "../pclhip.h"

namespace pclhip {

using index_t = std::int32_t;           // common/include/pcl/types.h:110-133
using Indices = std::vector<index_t>;

struct alignas(16) PointXYZ {
  float x = 0, y = 0, z = 0, w = 1.0f;  // data[4], data[3] = 1 (point_types.hpp:205-213)
  PointXYZ() = default;
  PointXYZ(float x_, float y_, float z_) : x(x_), y(y_), z(z_), w(1.0f) {}
};
struct alignas(16) Normal {
  float normal_x = 0, normal_y = 0, normal_z = 0, pad0 = 0;
  float curvature = 0, pad1[3] = {0, 0, 0};
};
struct alignas(16) PointNormal {
  float x = 0, y = 0, z = 0, w = 1.0f;
  float normal_x = 0, normal_y = 0, normal_z = 0, pad0 = 0;
  float curvature = 0, pad1[3] = {0, 0, 0};
};
static_assert(sizeof(PointXYZ) == 16 && sizeof(Normal) == 32 && sizeof(PointNormal) == 48, "PCL record sizes");

template <typename PointT>
struct PointCloud {
  using Ptr = std::shared_ptr<PointCloud<PointT>>;
  using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
  std::vector<PointT> points;
  std::uint32_t width = 0, height = 1;
  bool is_dense = true;
  // acquisition pose (common/include/pcl/point_cloud.h:406-408): origin x y z 0, orientation w x y z;
  // filled from a PCD file's VIEWPOINT line and used as NormalEstimation's default viewpoint
  float sensor_origin_[4] = {0, 0, 0, 0};
  float sensor_orientation_[4] = {1, 0, 0, 0};
  std::size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void resize(std::size_t n) { points.resize(n); width = std::uint32_t(n); height = 1; }
  void push_back(const PointT& p) { points.push_back(p); width = std::uint32_t(points.size()); }
  PointT& operator[](std::size_t i) { return points[i]; }
  const PointT& operator[](std::size_t i) const { return points[i]; }
  Ptr makeShared() const { return std::make_shared<PointCloud<PointT>>(*this); }
};

struct Correspondence {
  index_t index_query = 0, index_match = -1;
  float distance = FLT_MAX;  // squared (correspondence.h:66-71)
};
using Correspondences = std::vector<Correspondence>;

struct Matrix4f {  // row-major 4x4 (PCL hands out Eigen::Matrix4f; coefficient access is (row, col))
  float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float& operator()(int r, int c) { return m[4 * r + c]; }
  float operator()(int r, int c) const { return m[4 * r + c]; }
  static Matrix4f Identity() { return Matrix4f(); }
};

// One context per device, shared by the objects below (like PCL objects share nothing but the clouds).
class Context {
 public:
  explicit Context(int device = 0, void* hip_stream = nullptr) {
    if (pclhip_ctx_create(device, hip_stream, &ctx_) != PCLHIP_OK) {
      error_ = pclhip_last_error(nullptr);
      ctx_ = nullptr;
    }
  }
  ~Context() { if (ctx_) pclhip_ctx_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  bool ok() const { return ctx_ != nullptr; }
  pclhip_ctx* get() const { return ctx_; }
  std::string getLastError() const { return ctx_ ? pclhip_last_error(ctx_) : error_; }
  using Ptr = std::shared_ptr<Context>;
 private:
  pclhip_ctx* ctx_ = nullptr;
  std::string error_;
};

namespace search {

// pcl::search::KdTree<PointT>: setInputCloud / nearestKSearch (single + batch overloads).
template <typename PointT>
class KdTree {
 public:
  using Ptr = std::shared_ptr<KdTree<PointT>>;
  using PointCloudConstPtr = typename PointCloud<PointT>::ConstPtr;
  explicit KdTree(Context::Ptr ctx, bool sorted = true) : ctx_(std::move(ctx)) { (void)sorted; }
  ~KdTree() { if (index_) pclhip_index_destroy(index_); }
  KdTree(const KdTree&) = delete;
  KdTree& operator=(const KdTree&) = delete;

  // search/include/pcl/search/impl/kdtree.hpp:87-97; a repeated call with the same cloud object is
  // a no-op (the reference rebuilds twice on a cold align())
  bool setInputCloud(const PointCloudConstPtr& cloud, const std::shared_ptr<const Indices>& indices = nullptr) {
    if (!ctx_ || !ctx_->ok() || !cloud) return false;
    if (index_ && cloud == input_ && indices == indices_) return true;
    if (index_) { pclhip_index_destroy(index_); index_ = nullptr; }
    input_ = cloud;
    indices_ = indices;
    const pclhip_status st = pclhip_index_build(ctx_->get(), cloud->points.data(), sizeof(PointT), cloud->size(),
                                                indices ? indices->data() : nullptr, indices ? indices->size() : 0,
                                                &index_);
    return st == PCLHIP_OK;
  }
  PointCloudConstPtr getInputCloud() const { return input_; }

  // kdtree_flann.hpp:234-274: returns the number of neighbours found, resizes the outputs
  int nearestKSearch(const PointT& point, int k, Indices& k_indices, std::vector<float>& k_sqr_distances) const {
    if (!index_ || k < 1) return 0;
    const std::uint64_t n = pclhip_index_size(index_);
    if (std::uint64_t(k) > n) k = int(n);  // :241-242
    k_indices.resize(k);
    k_sqr_distances.resize(k);
    if (k == 0) return 0;
    if (pclhip_knn(index_, &point, sizeof(PointT), 1, k, k_indices.data(), k_sqr_distances.data()) != PCLHIP_OK) return 0;
    int found = 0;
    while (found < k && k_indices[found] >= 0) ++found;
    k_indices.resize(found);
    k_sqr_distances.resize(found);
    return found;
  }
  // batch overload, search/include/pcl/search/search.h:216-219 -- the efficient entry: one launch
  void nearestKSearch(const PointCloud<PointT>& cloud, const Indices& indices, int k, std::vector<Indices>& k_indices,
                      std::vector<std::vector<float>>& k_sqr_distances) const {
    k_indices.clear();
    k_sqr_distances.clear();
    if (!index_ || k < 1) return;
    std::vector<PointT> q;
    const PointT* qp = cloud.points.data();
    std::size_t nq = cloud.size();
    if (!indices.empty()) {
      q.reserve(indices.size());
      for (index_t i : indices) q.push_back(cloud[i]);
      qp = q.data();
      nq = q.size();
    }
    const std::uint64_t n = pclhip_index_size(index_);
    const int kk = std::uint64_t(k) > n ? int(n) : k;
    k_indices.assign(nq, Indices());
    k_sqr_distances.assign(nq, std::vector<float>());
    if (kk == 0 || nq == 0) return;
    Indices flat_i(nq * kk);
    std::vector<float> flat_d(nq * kk);
    if (pclhip_knn(index_, qp, sizeof(PointT), nq, kk, flat_i.data(), flat_d.data()) != PCLHIP_OK) return;
    for (std::size_t i = 0; i < nq; ++i) {
      int found = 0;
      while (found < kk && flat_i[i * kk + found] >= 0) ++found;
      k_indices[i].assign(flat_i.begin() + i * kk, flat_i.begin() + i * kk + found);
      k_sqr_distances[i].assign(flat_d.begin() + i * kk, flat_d.begin() + i * kk + found);
    }
  }
  // kdtree_flann.hpp:372-414: neighbours with squared distance < radius^2, ascending; max_nn = 0: all
  int radiusSearch(const PointT& point, double radius, Indices& k_indices, std::vector<float>& k_sqr_distances,
                   unsigned int max_nn = 0) const {
    k_indices.clear();
    k_sqr_distances.clear();
    if (!index_) return 0;
    std::uint64_t off[2] = {0, 0}, total = 0;
    pclhip_status st = pclhip_radius_search(index_, &point, sizeof(PointT), 1, radius, max_nn, off, nullptr, nullptr, 0, &total);
    if ((st != PCLHIP_OK && st != PCLHIP_ERR_OVERFLOW) || total == 0) return 0;
    k_indices.resize(total);
    k_sqr_distances.resize(total);
    if (pclhip_radius_search(index_, &point, sizeof(PointT), 1, radius, max_nn, off, k_indices.data(),
                             k_sqr_distances.data(), total, &total) != PCLHIP_OK) {
      k_indices.clear();
      k_sqr_distances.clear();
      return 0;
    }
    return int(total);
  }
  // batch overload (search/include/pcl/search/impl/search.hpp:164-190): one call for the whole cloud
  void radiusSearch(const PointCloud<PointT>& cloud, const Indices& indices, double radius,
                    std::vector<Indices>& k_indices, std::vector<std::vector<float>>& k_sqr_distances,
                    unsigned int max_nn = 0) const {
    k_indices.clear();
    k_sqr_distances.clear();
    if (!index_) return;
    std::vector<PointT> q;
    const PointT* qp = cloud.points.data();
    std::size_t nq = cloud.size();
    if (!indices.empty()) {
      for (index_t i : indices) q.push_back(cloud[i]);
      qp = q.data();
      nq = q.size();
    }
    k_indices.assign(nq, Indices());
    k_sqr_distances.assign(nq, std::vector<float>());
    if (nq == 0) return;
    std::vector<std::uint64_t> off(nq + 1, 0);
    std::uint64_t total = 0;
    pclhip_status st = pclhip_radius_search(index_, qp, sizeof(PointT), nq, radius, max_nn, off.data(), nullptr, nullptr, 0, &total);
    if ((st != PCLHIP_OK && st != PCLHIP_ERR_OVERFLOW) || total == 0) return;
    Indices flat_i(total);
    std::vector<float> flat_d(total);
    if (pclhip_radius_search(index_, qp, sizeof(PointT), nq, radius, max_nn, off.data(), flat_i.data(), flat_d.data(),
                             total, &total) != PCLHIP_OK) return;
    for (std::size_t i = 0; i < nq; ++i) {
      k_indices[i].assign(flat_i.begin() + off[i], flat_i.begin() + off[i + 1]);
      k_sqr_distances[i].assign(flat_d.begin() + off[i], flat_d.begin() + off[i + 1]);
    }
  }
  pclhip_index* handle() const { return index_; }
  Context::Ptr context() const { return ctx_; }

 private:
  Context::Ptr ctx_;
  pclhip_index* index_ = nullptr;
  PointCloudConstPtr input_;
  std::shared_ptr<const Indices> indices_;
};

}  // namespace search

// pcl::NormalEstimation<PointInT, pcl::Normal> in k-NN mode (setKSearch).
template <typename PointInT>
class NormalEstimation {
 public:
  explicit NormalEstimation(Context::Ptr ctx) : ctx_(std::move(ctx)) {}
  void setInputCloud(const typename PointCloud<PointInT>::ConstPtr& cloud) { input_ = cloud; }
  void setSearchMethod(const typename search::KdTree<PointInT>::Ptr& tree) { tree_ = tree; }
  void setKSearch(int k) { k_ = k; }
  void setRadiusSearch(double radius) { radius_ = radius; }
  // normal_3d.h:255-262 / :328-351: the cloud's sensor origin is the viewpoint until setViewPoint is called
  void setViewPoint(float x, float y, float z) { vp_[0] = x; vp_[1] = y; vp_[2] = z; use_sensor_origin_ = false; }
  void useSensorOriginAsViewPoint() { use_sensor_origin_ = true; }
  // Feature::compute (features/include/pcl/features/impl/feature.hpp:195-229); initCompute refuses
  // "both radius and K defined" and "neither defined" (:131-174)
  void compute(PointCloud<Normal>& output) {
    output.points.clear();
    if (!input_ || (k_ < 1) == !(radius_ > 0.0)) return;
    if (!tree_) tree_ = std::make_shared<search::KdTree<PointInT>>(ctx_);
    if (!tree_->setInputCloud(input_)) return;
    if (use_sensor_origin_) {
      vp_[0] = input_->sensor_origin_[0]; vp_[1] = input_->sensor_origin_[1]; vp_[2] = input_->sensor_origin_[2];
    }
    output.resize(input_->size());
    std::uint64_t nan = 0;
    std::vector<float> tmp(input_->size() * 4);
    const pclhip_status st = (k_ >= 1) ? pclhip_normals(tree_->handle(), k_, vp_, tmp.data(), 16, &nan)
                                       : pclhip_normals_radius(tree_->handle(), radius_, vp_, tmp.data(), 16, &nan);
    if (st != PCLHIP_OK) { output.points.clear(); return; }
    for (std::size_t i = 0; i < output.size(); ++i) {
      output[i].normal_x = tmp[4 * i]; output[i].normal_y = tmp[4 * i + 1]; output[i].normal_z = tmp[4 * i + 2];
      output[i].curvature = tmp[4 * i + 3];
    }
    output.is_dense = (nan == 0);  // normal_3d.hpp:56,63
  }
  typename search::KdTree<PointInT>::Ptr getSearchMethod() const { return tree_; }
 private:
  Context::Ptr ctx_;
  typename PointCloud<PointInT>::ConstPtr input_;
  typename search::KdTree<PointInT>::Ptr tree_;
  int k_ = 0;
  double radius_ = 0.0;
  float vp_[3] = {0, 0, 0};
  bool use_sensor_origin_ = true;
};

namespace registration {

// pcl::registration::CorrespondenceRejector{Distance,MedianDistance,OneToOne,Trimmed}: parameter holders;
// the rejection itself runs on the device inside the ICP iteration (pclhip_icp_set_rejectors).
struct CorrespondenceRejector {
  using Ptr = std::shared_ptr<CorrespondenceRejector>;
  pclhip_rejector desc{PCLHIP_REJ_DISTANCE, 0.0, 0, 0};
  virtual ~CorrespondenceRejector() = default;
};
struct CorrespondenceRejectorDistance : CorrespondenceRejector {
  CorrespondenceRejectorDistance() { desc.kind = PCLHIP_REJ_DISTANCE; }
  void setMaximumDistance(float d) { desc.param = d; }  // correspondence_rejection_distance.h:93-97
};
struct CorrespondenceRejectorMedianDistance : CorrespondenceRejector {
  CorrespondenceRejectorMedianDistance() { desc.kind = PCLHIP_REJ_MEDIAN_DISTANCE; desc.param = 1.0; }
  void setMedianFactor(double f) { desc.param = f; }
};
struct CorrespondenceRejectorOneToOne : CorrespondenceRejector {
  CorrespondenceRejectorOneToOne() { desc.kind = PCLHIP_REJ_ONE_TO_ONE; }
};
struct CorrespondenceRejectorTrimmed : CorrespondenceRejector {
  CorrespondenceRejectorTrimmed() { desc.kind = PCLHIP_REJ_TRIMMED; desc.param = 0.5; }
  void setOverlapRatio(float r) { desc.param = r; }
  void setMinCorrespondences(unsigned n) { desc.min_correspondences = n; }
};

// pcl::registration::CorrespondenceEstimation::determineCorrespondences
template <typename PointSource, typename PointTarget>
class CorrespondenceEstimation {
 public:
  explicit CorrespondenceEstimation(Context::Ptr ctx) : ctx_(std::move(ctx)) {}
  void setInputSource(const typename PointCloud<PointSource>::ConstPtr& c) { source_ = c; }
  void setInputTarget(const typename PointCloud<PointTarget>::ConstPtr& c) { target_ = c; }
  void setSearchMethodTarget(const typename search::KdTree<PointTarget>::Ptr& t, bool = false) { tree_ = t; }
  void determineReciprocalCorrespondences(Correspondences& out, double max_distance = std::sqrt(DBL_MAX)) {
    reciprocal_ = true;
    determineCorrespondences(out, max_distance);
    reciprocal_ = false;
  }
  void determineCorrespondences(Correspondences& out, double max_distance = std::sqrt(DBL_MAX)) {
    out.clear();
    if (!source_ || !target_) return;
    if (!tree_) tree_ = std::make_shared<search::KdTree<PointTarget>>(ctx_);
    if (!tree_->setInputCloud(target_)) return;
    pclhip_icp* icp = nullptr;
    if (pclhip_icp_create(tree_->handle(), &icp) != PCLHIP_OK) return;
    const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double sums[PCLHIP_ICP_NSUMS];
    if (pclhip_icp_set_source(icp, source_->points.data(), sizeof(PointSource), source_->size()) == PCLHIP_OK &&
        pclhip_icp_set_reciprocal(icp, reciprocal_ ? 1 : 0) == PCLHIP_OK &&
        pclhip_icp_iterate(icp, I, max_distance, PCLHIP_ICP_POINT_TO_POINT, sums) == PCLHIP_OK) {
      const std::size_t n = source_->size();
      Indices q(n), m(n);
      std::vector<float> d(n);
      std::uint64_t cnt = 0;
      if (pclhip_icp_fetch_correspondences(icp, q.data(), m.data(), d.data(), &cnt) == PCLHIP_OK) {
        out.resize(cnt);
        for (std::uint64_t i = 0; i < cnt; ++i) { out[i].index_query = q[i]; out[i].index_match = m[i]; out[i].distance = d[i]; }
      }
    }
    pclhip_icp_destroy(icp);
  }
 private:
  Context::Ptr ctx_;
  typename PointCloud<PointSource>::ConstPtr source_;
  typename PointCloud<PointTarget>::ConstPtr target_;
  typename search::KdTree<PointTarget>::Ptr tree_;
  bool reciprocal_ = false;
};

}  // namespace registration

// pcl::IterativeClosestPoint<PointSource, PointTarget>; MODE selects the estimator exactly as the
// reference's two classes do (icp.h:149-151 SVD, icp.h:395-398 point-to-plane LLS).
template <typename PointSource, typename PointTarget, int MODE = PCLHIP_ICP_POINT_TO_POINT>
class IterativeClosestPoint {
 public:
  using PointCloudSource = PointCloud<PointSource>;
  using PointCloudTarget = PointCloud<PointTarget>;
  explicit IterativeClosestPoint(Context::Ptr ctx) : ctx_(std::move(ctx)) { pclhip_icp_params_default(&p_); p_.mode = MODE; }
  ~IterativeClosestPoint() { if (icp_) pclhip_icp_destroy(icp_); }
  IterativeClosestPoint(const IterativeClosestPoint&) = delete;             // icp.h:168-173
  IterativeClosestPoint& operator=(const IterativeClosestPoint&) = delete;

  void setInputSource(const typename PointCloudSource::ConstPtr& c) { source_ = c; source_dirty_ = true; }
  void setInputTarget(const typename PointCloudTarget::ConstPtr& c) { target_ = c; target_dirty_ = true; }
  void setSearchMethodTarget(const typename search::KdTree<PointTarget>::Ptr& t, bool force_no_recompute = false) {
    tree_ = t; force_no_recompute_ = force_no_recompute; target_dirty_ = true;
  }
  void setMaximumIterations(int n) { p_.max_iterations = n; }
  void setMaxCorrespondenceDistance(double d) { p_.max_correspondence_distance = d; }
  void setTransformationEpsilon(double e) { p_.transformation_epsilon = e; }
  void setTransformationRotationEpsilon(double e) { p_.transformation_rotation_epsilon = e; }
  void setEuclideanFitnessEpsilon(double e) { p_.euclidean_fitness_epsilon = e; }
  // Registration::addCorrespondenceRejector (registration.h:430-434), icp.h:251-256
  void addCorrespondenceRejector(const registration::CorrespondenceRejector::Ptr& r) { rejectors_.push_back(r); filters_dirty_ = true; }
  void clearCorrespondenceRejectors() { rejectors_.clear(); filters_dirty_ = true; }
  void setUseReciprocalCorrespondences(bool on) { reciprocal_ = on; filters_dirty_ = true; }
  // IterativeClosestPointWithNormals::setUseSymmetricObjective / setEnforceSameDirectionNormals
  // (icp.h:380-428): TransformationEstimationSymmetricPointToPlaneLLS; source AND target need normals
  void setUseSymmetricObjective(bool on) {
    static_assert(MODE == PCLHIP_ICP_POINT_TO_PLANE, "only IterativeClosestPointWithNormals has this option");
    p_.mode = on ? PCLHIP_ICP_SYMMETRIC : PCLHIP_ICP_POINT_TO_PLANE;
  }
  bool getUseSymmetricObjective() const { return p_.mode == PCLHIP_ICP_SYMMETRIC; }
  void setEnforceSameDirectionNormals(bool on) { enforce_same_direction_ = on; filters_dirty_ = true; }
  bool getEnforceSameDirectionNormals() const { return enforce_same_direction_; }
  int getMaximumIterations() const { return p_.max_iterations; }
  double getMaxCorrespondenceDistance() const { return p_.max_correspondence_distance; }

  // Registration::align (registration/include/pcl/registration/impl/registration.hpp:170-221)
  void align(PointCloudSource& output, const Matrix4f& guess = Matrix4f::Identity()) {
    converged_ = false;
    if (!initCompute()) return;
    pclhip_icp_result r;
    if (pclhip_icp_align(icp_, &p_, guess.m, &r) != PCLHIP_OK) return;
    std::memcpy(final_.m, r.final_transformation, sizeof final_.m);
    converged_ = r.converged != 0;
    nr_iterations_ = r.nr_iterations;
    state_ = r.convergence_state;
    last_mse_ = r.mse;
    output = *source_;  // icp.hpp:264-267: copy all fields, then transform xyz (+ normals)
    const std::size_t nrm_off = (MODE == PCLHIP_ICP_POINT_TO_PLANE && sizeof(PointSource) >= 28) ? 16 : 0;
    pclhip_transform_cloud(ctx_->get(), final_.m, MODE == PCLHIP_ICP_POINT_TO_PLANE ? 1 : 0, source_->points.data(),
                           output.points.data(), sizeof(PointSource), source_->size(), nrm_off);
  }
  Matrix4f getFinalTransformation() const { return final_; }
  bool hasConverged() const { return converged_; }
  // Registration::getFitnessScore (registration/include/pcl/registration/impl/registration.hpp:132-168)
  double getFitnessScore(double max_range = std::numeric_limits<double>::max()) {
    if (!initCompute()) return std::numeric_limits<double>::max();
    double score = std::numeric_limits<double>::max();
    pclhip_icp_fitness_score(icp_, final_.m, max_range, &score, nullptr);
    return score;
  }
  int getNumberOfIterations() const { return nr_iterations_; }
  int getConvergenceState() const { return state_; }
  double getLastMSE() const { return last_mse_; }
  std::string getLastError() const { return ctx_->getLastError(); }

 private:
  // Registration::initCompute (impl/registration.hpp:73-101): (re)build the target tree only when
  // the target changed and force_no_recompute was not requested
  bool initCompute() {
    if (!ctx_ || !ctx_->ok() || !source_ || (!target_ && !tree_)) return false;
    if (!tree_) tree_ = std::make_shared<search::KdTree<PointTarget>>(ctx_);
    if (target_dirty_) {
      if (!(force_no_recompute_ && tree_->handle())) {
        if (!target_ || !tree_->setInputCloud(target_)) return false;
      }
      if (MODE == PCLHIP_ICP_POINT_TO_PLANE && target_ && sizeof(PointTarget) >= 28) {
        // pcl::PointNormal target: normals live at +16 in every 48-byte record
        const char* base = reinterpret_cast<const char*>(target_->points.data());
        if (pclhip_index_set_normals(tree_->handle(), base + 16, sizeof(PointTarget)) != PCLHIP_OK) return false;
      }
      if (icp_) { pclhip_icp_destroy(icp_); icp_ = nullptr; }
      target_dirty_ = false;
      source_dirty_ = true;
    }
    if (!icp_ && pclhip_icp_create(tree_->handle(), &icp_) != PCLHIP_OK) return false;
    if (source_dirty_) {
      if (pclhip_icp_set_source(icp_, source_->points.data(), sizeof(PointSource), source_->size()) != PCLHIP_OK) return false;
      if (MODE == PCLHIP_ICP_POINT_TO_PLANE && sizeof(PointSource) >= 28) {  // pcl::PointNormal source
        const char* base = reinterpret_cast<const char*>(source_->points.data());
        if (pclhip_icp_set_source_normals(icp_, base + 16, sizeof(PointSource)) != PCLHIP_OK) return false;
      }
      source_dirty_ = false;
      filters_dirty_ = true;
    }
    if (filters_dirty_) {
      std::vector<pclhip_rejector> list;
      for (const auto& r : rejectors_) list.push_back(r->desc);
      if (pclhip_icp_set_rejectors(icp_, list.data(), int(list.size())) != PCLHIP_OK) return false;
      if (pclhip_icp_set_reciprocal(icp_, reciprocal_ ? 1 : 0) != PCLHIP_OK) return false;
      if (pclhip_icp_set_enforce_same_direction_normals(icp_, enforce_same_direction_ ? 1 : 0) != PCLHIP_OK) return false;
      filters_dirty_ = false;
    }
    return true;
  }
  Context::Ptr ctx_;
  pclhip_icp_params p_;
  pclhip_icp* icp_ = nullptr;
  typename PointCloudSource::ConstPtr source_;
  typename PointCloudTarget::ConstPtr target_;
  typename search::KdTree<PointTarget>::Ptr tree_;
  bool force_no_recompute_ = false, target_dirty_ = true, source_dirty_ = true, filters_dirty_ = true;
  bool reciprocal_ = false, enforce_same_direction_ = true;
  std::vector<registration::CorrespondenceRejector::Ptr> rejectors_;
  Matrix4f final_;
  bool converged_ = false;
  int nr_iterations_ = 0, state_ = 0;
  double last_mse_ = 0;
};

template <typename PointSource, typename PointTarget>
using IterativeClosestPointWithNormals = IterativeClosestPoint<PointSource, PointTarget, PCLHIP_ICP_POINT_TO_PLANE>;

namespace registration {
// TransformationEstimationSVD / PointToPlaneLLS / SymmetricPointToPlaneLLS::estimateRigidTransformation
// (cloud_src, cloud_tgt, T): pair i = (src[i], tgt[i]) (transformation_estimation.h:71-115).  Normals are
// read from the PointNormal layout (+16 bytes).
template <typename PointSource, typename PointTarget, int MODE>
class TransformationEstimation {
 public:
  explicit TransformationEstimation(Context::Ptr ctx) : ctx_(std::move(ctx)) {}
  void setEnforceSameDirectionNormals(bool on) { enforce_ = on; }
  bool estimateRigidTransformation(const PointCloud<PointSource>& src, const PointCloud<PointTarget>& tgt,
                                   Matrix4f& T) const {
    if (src.size() != tgt.size()) return false;  // "Number or points in source differs than target"
    const char* s = reinterpret_cast<const char*>(src.points.data());
    const char* t = reinterpret_cast<const char*>(tgt.points.data());
    const void* sn = (MODE == PCLHIP_ICP_SYMMETRIC && sizeof(PointSource) >= 28) ? s + 16 : nullptr;
    const void* tn = (MODE != PCLHIP_ICP_POINT_TO_POINT && sizeof(PointTarget) >= 28) ? t + 16 : nullptr;
    return pclhip_estimate_rigid_transformation(ctx_->get(), MODE, s, sizeof(PointSource), sn, sizeof(PointSource), t,
                                                sizeof(PointTarget), tn, sizeof(PointTarget), src.size(),
                                                enforce_ ? 1 : 0, T.m, nullptr) == PCLHIP_OK;
  }
 private:
  Context::Ptr ctx_;
  bool enforce_ = true;
};
template <typename S, typename T> using TransformationEstimationSVD = TransformationEstimation<S, T, PCLHIP_ICP_POINT_TO_POINT>;
template <typename S, typename T> using TransformationEstimationPointToPlaneLLS = TransformationEstimation<S, T, PCLHIP_ICP_POINT_TO_PLANE>;
template <typename S, typename T> using TransformationEstimationSymmetricPointToPlaneLLS = TransformationEstimation<S, T, PCLHIP_ICP_SYMMETRIC>;
}  // namespace registration

// pcl::io::loadPCDFile / savePCDFile{ASCII,Binary,BinaryCompressed} (io/include/pcl/io/pcd_io.h:685-800)
// for the point types of this header: records of sizeof(PointT) bytes, normals at +16 when the type has them.
namespace io {
template <typename PointT>
int loadPCDFile(const std::string& file_name, PointCloud<PointT>& cloud) {
  pclhip_pcd_info info;
  if (pclhip_pcd_read_header(file_name.c_str(), &info) != PCLHIP_OK) return -1;
  cloud.points.assign(info.points, PointT());
  uint64_t n = 0;
  int dense = 1;
  const std::size_t nrm_off = sizeof(PointT) >= 28 ? 16 : 0;
  if (pclhip_pcd_read(file_name.c_str(), cloud.points.data(), sizeof(PointT), nrm_off, info.points, &n, &dense) != PCLHIP_OK)
    return -1;
  cloud.width = info.width;
  cloud.height = info.height;
  cloud.is_dense = dense != 0;
  for (int i = 0; i < 3; ++i) cloud.sensor_origin_[i] = info.viewpoint[i];   // VIEWPOINT tx ty tz qw qx qy qz
  cloud.sensor_origin_[3] = 0.0f;
  for (int i = 0; i < 4; ++i) cloud.sensor_orientation_[i] = info.viewpoint[3 + i];
  return 0;
}
template <typename PointT>
int savePCDFile(const std::string& file_name, const PointCloud<PointT>& cloud, int data_type, int precision = 8) {
  const std::size_t nrm_off = sizeof(PointT) >= 28 ? 16 : 0;
  // width, height and the acquisition pose travel with the cloud (PCDWriter::generateHeader, pcd_io.cpp:848-1043)
  const float vp[7] = {cloud.sensor_origin_[0], cloud.sensor_origin_[1], cloud.sensor_origin_[2],
                       cloud.sensor_orientation_[0], cloud.sensor_orientation_[1], cloud.sensor_orientation_[2],
                       cloud.sensor_orientation_[3]};
  const bool organized = std::uint64_t(cloud.width) * cloud.height == cloud.size();
  const std::uint32_t w = organized ? cloud.width : std::uint32_t(cloud.size()), h = organized ? cloud.height : 1;
  return pclhip_pcd_write_organized(file_name.c_str(), cloud.points.data(), sizeof(PointT), nrm_off, w, h, vp, data_type,
                                    precision) == PCLHIP_OK ? 0 : -1;
}
template <typename PointT> int savePCDFileASCII(const std::string& f, const PointCloud<PointT>& c) { return savePCDFile(f, c, 0); }
template <typename PointT> int savePCDFileBinary(const std::string& f, const PointCloud<PointT>& c) { return savePCDFile(f, c, 1); }
template <typename PointT> int savePCDFileBinaryCompressed(const std::string& f, const PointCloud<PointT>& c) { return savePCDFile(f, c, 2); }
}  // namespace io

// pcl::VoxelGrid<pcl::PointXYZ>
class VoxelGrid {
 public:
  explicit VoxelGrid(Context::Ptr ctx) : ctx_(std::move(ctx)) {}
  void setInputCloud(const PointCloud<PointXYZ>::ConstPtr& c) { input_ = c; }
  void setLeafSize(float lx, float ly, float lz) { leaf_[0] = lx; leaf_[1] = ly; leaf_[2] = lz; }
  void setMinimumPointsNumberPerVoxel(unsigned n) { min_pts_ = n; }
  void setFilterFieldName(const std::string& f) { field_ = f; }
  void setFilterLimits(double lo, double hi) { lo_ = lo; hi_ = hi; }
  // Filter::filter -> applyFilter (filters/include/pcl/filters/impl/voxel_grid.hpp:597-814); when the
  // voxel index would overflow the reference warns and returns the input unchanged (:620-629)
  void filter(PointCloud<PointXYZ>& output) {
    output.points.clear();
    output.height = 1;
    output.is_dense = true;
    if (!input_) { output.width = 0; return; }
    std::vector<PointXYZ> out(input_->size());
    std::uint64_t n = 0;
    const pclhip_status st = pclhip_voxelgrid(ctx_->get(), input_->points.data(), sizeof(PointXYZ), input_->size(), leaf_,
                                              min_pts_, field_ == "z", lo_, hi_, out.data(), &n);
    if (st == PCLHIP_ERR_OVERFLOW) { output = *input_; return; }
    if (st != PCLHIP_OK) { output.width = 0; return; }
    out.resize(n);
    output.points.swap(out);
    output.width = std::uint32_t(n);
  }
 private:
  Context::Ptr ctx_;
  PointCloud<PointXYZ>::ConstPtr input_;
  float leaf_[3] = {0, 0, 0};
  unsigned min_pts_ = 0;
  std::string field_;
  double lo_ = -FLT_MAX, hi_ = FLT_MAX;
};

}  // namespace pclhip
