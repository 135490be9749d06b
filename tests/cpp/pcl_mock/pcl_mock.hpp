// pcl_mock.hpp -- TEST INFRASTRUCTURE: a minimal, Eigen-free stand-in for the PCL base classes that
// include/pclhip/pcl_plugin.hpp derives from, so that the real-PCL binding compiles and runs here (PCL itself
// needs Eigen, Boost and FLANN; none is installed).  Only the members the binding touches exist; every
// signature is transcribed from the PCL header named next to it (paths relative to the PCL tree), the CPU
// implementations behind them are NOT reproduced (the mock aborts where PCL would run its CPU path).
// The forwarding headers next to this file give it PCL's include paths (<pcl/search/kdtree.h> ...).
#pragma once

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <limits>
#include <memory>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace Eigen {  // just enough of Eigen::Matrix for Matrix4f / Vector4f / Quaternionf members
template <typename S, int R, int C>
struct Matrix {
  S d[R * C] = {};
  S& operator()(int r, int c) { return d[r * C + c]; }
  const S& operator()(int r, int c) const { return d[r * C + c]; }
  S& operator[](int i) { return d[i]; }
  const S& operator[](int i) const { return d[i]; }
  S coeff(int i) const { return d[i]; }
  static Matrix Identity() {
    Matrix m;
    for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = S(1);
    return m;
  }
  static Matrix Zero() { return Matrix(); }
  void setIdentity() { *this = Identity(); }
  Matrix operator*(const Matrix& o) const {
    static_assert(R == C, "square only");
    Matrix r;
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) {
        S s = 0;
        for (int k = 0; k < C; ++k) s += (*this)(i, k) * o(k, j);
        r(i, j) = s;
      }
    return r;
  }
};
using Matrix4f = Matrix<float, 4, 4>;
using Vector4f = Matrix<float, 4, 1>;
using Vector3f = Matrix<float, 3, 1>;
using Vector4i = Matrix<int, 4, 1>;
using Vector3i = Matrix<int, 3, 1>;
using Array4f = Matrix<float, 4, 1>;
struct Quaternionf {
  float w_ = 1, x_ = 0, y_ = 0, z_ = 0;
  static Quaternionf Identity() { return Quaternionf(); }
};
}  // namespace Eigen

namespace pcl {

template <typename T> using shared_ptr = std::shared_ptr<T>;          // common/include/pcl/memory.h
using index_t = std::int32_t;                                          // common/include/pcl/types.h:110-133
using Indices = std::vector<index_t>;
using IndicesPtr = shared_ptr<Indices>;
using IndicesConstPtr = shared_ptr<const Indices>;

[[noreturn]] inline void mock_no_cpu_path(const char* what) {
  std::fprintf(stderr, "pcl_mock: %s is PCL's CPU implementation, which the mock does not contain\n", what);
  std::abort();
}

// common/include/pcl/impl/point_types.hpp:315-321, 787-794, 843-853
struct alignas(16) PointXYZ {
  union { float data[4]; struct { float x, y, z; }; };
  PointXYZ() : data{0, 0, 0, 1.0f} {}
  PointXYZ(float x_, float y_, float z_) : data{x_, y_, z_, 1.0f} {}
};
struct alignas(16) Normal {
  union { float data_n[4]; float normal[3]; struct { float normal_x, normal_y, normal_z; }; };
  union { struct { float curvature; }; float data_c[4]; };
  Normal() : data_n{0, 0, 0, 0}, data_c{0, 0, 0, 0} {}
};
struct alignas(16) PointNormal {
  union { float data[4]; struct { float x, y, z; }; };
  union { float data_n[4]; float normal[3]; struct { float normal_x, normal_y, normal_z; }; };
  union { struct { float curvature; }; float data_c[4]; };
  PointNormal() : data{0, 0, 0, 1.0f}, data_n{0, 0, 0, 0}, data_c{0, 0, 0, 0} {}
};
static_assert(sizeof(PointXYZ) == 16 && sizeof(Normal) == 32 && sizeof(PointNormal) == 48, "PCL record sizes");
// two layouts the device path must NOT take for PointNormal's: impl/point_types.hpp:390-400 (intensity at +16) and
// :885-925 (rgb at +32, curvature at +36)
struct alignas(16) PointXYZI {
  union { float data[4]; struct { float x, y, z; }; };
  union { struct { float intensity; }; float data_c[4]; };
  PointXYZI() : data{0, 0, 0, 1.0f}, data_c{0, 0, 0, 0} {}
};
struct alignas(16) PointXYZRGBNormal {
  union { float data[4]; struct { float x, y, z; }; };
  union { float data_n[4]; float normal[3]; struct { float normal_x, normal_y, normal_z; }; };
  union { struct { float rgb; float curvature; }; float data_c[4]; };
  PointXYZRGBNormal() : data{0, 0, 0, 1.0f}, data_n{0, 0, 0, 0}, data_c{0, 0, 0, 0} {}
};
static_assert(sizeof(PointXYZI) == 32 && sizeof(PointXYZRGBNormal) == 48, "PCL record sizes");

struct PCLHeader { std::uint32_t seq = 0; std::uint64_t stamp = 0; std::string frame_id; };

// common/include/pcl/point_cloud.h:173,393-409
template <typename PointT>
class PointCloud {
 public:
  using Ptr = shared_ptr<PointCloud<PointT>>;
  using ConstPtr = shared_ptr<const PointCloud<PointT>>;
  PCLHeader header;
  std::vector<PointT> points;
  std::uint32_t width = 0, height = 0;
  bool is_dense = true;
  Eigen::Vector4f sensor_origin_;
  Eigen::Quaternionf sensor_orientation_;
  std::size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void resize(std::size_t n) { points.resize(n); if (width * height != n) { width = std::uint32_t(n); height = 1; } }
  void push_back(const PointT& p) { points.push_back(p); width = std::uint32_t(points.size()); height = 1; }
  PointT& operator[](std::size_t i) { return points[i]; }
  const PointT& operator[](std::size_t i) const { return points[i]; }
  Ptr makeShared() const { return Ptr(new PointCloud<PointT>(*this)); }
};

// common/include/pcl/correspondence.h:60-91
struct Correspondence {
  index_t index_query = 0, index_match = -1;
  union { float distance; float weight; };
  Correspondence() : distance(std::numeric_limits<float>::max()) {}
  Correspondence(index_t q, index_t m, float d) : index_query(q), index_match(m), distance(d) {}
};
using Correspondences = std::vector<Correspondence>;
using CorrespondencesPtr = shared_ptr<Correspondences>;

// common/include/pcl/pcl_base.h:65-175
template <typename PointT>
class PCLBase {
 public:
  using PointCloud = pcl::PointCloud<PointT>;
  using PointCloudPtr = typename PointCloud::Ptr;
  using PointCloudConstPtr = typename PointCloud::ConstPtr;
  PCLBase() = default;
  virtual ~PCLBase() = default;
  virtual void setInputCloud(const PointCloudConstPtr& cloud) { input_ = cloud; }
  PointCloudConstPtr const getInputCloud() const { return input_; }
  virtual void setIndices(const IndicesPtr& indices) { indices_ = indices; fake_indices_ = false; use_indices_ = true; }
  virtual void setIndices(const IndicesConstPtr& indices) {
    indices_.reset(new Indices(*indices)); fake_indices_ = false; use_indices_ = true;
  }
  IndicesPtr getIndices() { return indices_; }
  IndicesConstPtr const getIndices() const { return indices_; }
 protected:
  PointCloudConstPtr input_;
  IndicesPtr indices_;
  bool use_indices_ = false, fake_indices_ = false;
  bool initCompute() {  // common/include/pcl/impl/pcl_base.hpp:138-174
    if (!input_) return false;
    if (!indices_) { fake_indices_ = true; indices_.reset(new Indices); }
    if (fake_indices_ && indices_->size() != input_->size()) {
      indices_->resize(input_->size());
      for (std::size_t i = 0; i < indices_->size(); ++i) (*indices_)[i] = index_t(i);
    }
    return true;
  }
  bool deinitCompute() { return true; }
};

// common/include/pcl/point_representation.h:59-190 (+ DefaultPointRepresentation<PointXYZ> :256-279,
// CustomPointRepresentation :546-579)
template <typename PointT>
class PointRepresentation {
 protected:
  int nr_dimensions_ = 0;
  std::vector<float> alpha_;
  bool trivial_ = false;
 public:
  using Ptr = shared_ptr<PointRepresentation<PointT>>;
  using ConstPtr = shared_ptr<const PointRepresentation<PointT>>;
  virtual ~PointRepresentation() = default;
  virtual void copyToFloatArray(const PointT& p, float* out) const = 0;
  bool isTrivial() const { return trivial_ && alpha_.empty(); }
  virtual bool isValid(const PointT& p) const {
    std::vector<float> t(nr_dimensions_);
    copyToFloatArray(p, t.data());
    for (float v : t) if (!std::isfinite(v)) return false;
    return true;
  }
  template <typename OutputType> void vectorize(const PointT& p, OutputType& out) const {
    std::vector<float> t(nr_dimensions_);
    copyToFloatArray(p, t.data());
    for (int i = 0; i < nr_dimensions_; ++i) out[i] = alpha_.empty() ? t[i] : t[i] * alpha_[i];
  }
  void setRescaleValues(const float* rescale_array) { alpha_.assign(rescale_array, rescale_array + nr_dimensions_); }
  int getNumberOfDimensions() const { return nr_dimensions_; }
};
template <typename PointT>
class DefaultPointRepresentation : public PointRepresentation<PointT> {
 public:
  DefaultPointRepresentation() { this->nr_dimensions_ = 3; this->trivial_ = true; }
  void copyToFloatArray(const PointT& p, float* out) const override { out[0] = p.x; out[1] = p.y; out[2] = p.z; }
};
template <typename PointT>
class CustomPointRepresentation : public PointRepresentation<PointT> {
 public:
  CustomPointRepresentation(int max_dim = 3, int start_dim = 0) : start_dim_(start_dim) { this->nr_dimensions_ = max_dim; }
  void copyToFloatArray(const PointT& p, float* out) const override {
    const float* f = reinterpret_cast<const float*>(&p) + start_dim_;
    for (int i = 0; i < this->nr_dimensions_; ++i) out[i] = f[i];
  }
 private:
  int start_dim_;
};

// kdtree/include/pcl/kdtree/kdtree_flann.h: only the type (the default Tree argument of search::KdTree)
template <typename PointT>
class KdTreeFLANN {
 public:
  using Ptr = shared_ptr<KdTreeFLANN<PointT>>;
  using ConstPtr = shared_ptr<const KdTreeFLANN<PointT>>;
};

namespace search {

// search/include/pcl/search/search.h:60-420
template <typename PointT>
class Search {
 public:
  using PointCloud = pcl::PointCloud<PointT>;
  using PointCloudPtr = typename PointCloud::Ptr;
  using PointCloudConstPtr = typename PointCloud::ConstPtr;
  using Ptr = shared_ptr<pcl::search::Search<PointT>>;
  using ConstPtr = shared_ptr<const pcl::search::Search<PointT>>;
  using IndicesPtr = pcl::IndicesPtr;
  using IndicesConstPtr = pcl::IndicesConstPtr;

  Search(const std::string& name = "", bool sorted = false) : sorted_results_(sorted), name_(name) {}
  virtual ~Search() = default;
  virtual const std::string& getName() const { return name_; }
  virtual void setSortedResults(bool sorted) { sorted_results_ = sorted; }
  virtual bool getSortedResults() { return sorted_results_; }
  virtual bool setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) {
    input_ = cloud; indices_ = indices; return true;
  }
  virtual PointCloudConstPtr getInputCloud() const { return input_; }
  virtual IndicesConstPtr getIndices() const { return indices_; }

  virtual int nearestKSearch(const PointT& point, int k, Indices& k_indices,
                             std::vector<float>& k_sqr_distances) const = 0;
  template <typename PointTDiff>
  int nearestKSearchT(const PointTDiff& point, int k, Indices& k_indices, std::vector<float>& k_sqr_distances) const {
    PointT p;
    p.x = point.x; p.y = point.y; p.z = point.z;  // copyPoint
    return nearestKSearch(p, k, k_indices, k_sqr_distances);
  }
  virtual int nearestKSearch(const PointCloud& cloud, index_t index, int k, Indices& k_indices,
                             std::vector<float>& k_sqr_distances) const {
    return nearestKSearch(cloud[index], k, k_indices, k_sqr_distances);
  }
  virtual int nearestKSearch(index_t index, int k, Indices& k_indices, std::vector<float>& k_sqr_distances) const {
    return nearestKSearch((*input_)[indices_ ? (*indices_)[index] : index], k, k_indices, k_sqr_distances);
  }
  virtual void nearestKSearch(const PointCloud& cloud, const Indices& indices, int k, std::vector<Indices>& k_indices,
                              std::vector<std::vector<float>>& k_sqr_distances) const {
    const std::size_t n = indices.empty() ? cloud.size() : indices.size();  // search.hpp:113-136
    k_indices.resize(n);
    k_sqr_distances.resize(n);
    for (std::size_t i = 0; i < n; ++i)
      nearestKSearch(cloud, indices.empty() ? index_t(i) : indices[i], k, k_indices[i], k_sqr_distances[i]);
  }
  virtual int radiusSearch(const PointT& point, double radius, Indices& k_indices,
                           std::vector<float>& k_sqr_distances, unsigned int max_nn = 0) const = 0;
  virtual int radiusSearch(const PointCloud& cloud, index_t index, double radius, Indices& k_indices,
                           std::vector<float>& k_sqr_distances, unsigned int max_nn = 0) const {
    return radiusSearch(cloud[index], radius, k_indices, k_sqr_distances, max_nn);
  }
  virtual void radiusSearch(const PointCloud& cloud, const Indices& indices, double radius,
                            std::vector<Indices>& k_indices, std::vector<std::vector<float>>& k_sqr_distances,
                            unsigned int max_nn = 0) const {
    const std::size_t n = indices.empty() ? cloud.size() : indices.size();  // search.hpp:164-190
    k_indices.resize(n);
    k_sqr_distances.resize(n);
    for (std::size_t i = 0; i < n; ++i)
      radiusSearch(cloud, indices.empty() ? index_t(i) : indices[i], radius, k_indices[i], k_sqr_distances[i], max_nn);
  }
 protected:
  PointCloudConstPtr input_;
  IndicesConstPtr indices_;
  bool sorted_results_;
  std::string name_;
};

// search/include/pcl/search/kdtree.h:61-168
template <typename PointT, class Tree = pcl::KdTreeFLANN<PointT>>
class KdTree : public Search<PointT> {
 public:
  using PointCloud = typename Search<PointT>::PointCloud;
  using PointCloudConstPtr = typename Search<PointT>::PointCloudConstPtr;
  using pcl::search::Search<PointT>::indices_;
  using pcl::search::Search<PointT>::input_;
  using pcl::search::Search<PointT>::nearestKSearch;
  using pcl::search::Search<PointT>::radiusSearch;
  using pcl::search::Search<PointT>::sorted_results_;
  using Ptr = shared_ptr<KdTree<PointT, Tree>>;
  using ConstPtr = shared_ptr<const KdTree<PointT, Tree>>;
  using KdTreePtr = typename Tree::Ptr;
  using PointRepresentationConstPtr = typename PointRepresentation<PointT>::ConstPtr;

  KdTree(bool sorted = true) : Search<PointT>("KdTree", sorted) { mock_no_cpu_path("pcl::KdTreeFLANN"); }
  ~KdTree() override = default;
  virtual void setPointRepresentation(const PointRepresentationConstPtr&) { mock_no_cpu_path("pcl::KdTreeFLANN"); }
  virtual PointRepresentationConstPtr getPointRepresentation() const { mock_no_cpu_path("pcl::KdTreeFLANN"); }
  void setSortedResults(bool sorted_results) override { sorted_results_ = sorted_results; }
  virtual void setEpsilon(float) { mock_no_cpu_path("pcl::KdTreeFLANN"); }
  virtual float getEpsilon() const { mock_no_cpu_path("pcl::KdTreeFLANN"); }
  bool setInputCloud(const PointCloudConstPtr&, const IndicesConstPtr& = IndicesConstPtr()) override {
    mock_no_cpu_path("pcl::KdTreeFLANN");
  }
  int nearestKSearch(const PointT&, int, Indices&, std::vector<float>&) const override { mock_no_cpu_path("pcl::KdTreeFLANN"); }
  int radiusSearch(const PointT&, double, Indices&, std::vector<float>&, unsigned int = 0) const override {
    mock_no_cpu_path("pcl::KdTreeFLANN");
  }
 protected:
  KdTree(const std::string& name, bool sorted) : Search<PointT>(name, sorted) {}  // leaves tree_ uninitialised (:165-167)
  KdTreePtr tree_;
};

}  // namespace search

namespace registration {

// registration/include/pcl/registration/correspondence_rejection.h:55-200 and the four rejectors of
// correspondence_rejection_{distance,median_distance,one_to_one,trimmed}.h (parameters only)
class CorrespondenceRejector {
 public:
  using Ptr = shared_ptr<CorrespondenceRejector>;
  using ConstPtr = shared_ptr<const CorrespondenceRejector>;
  virtual ~CorrespondenceRejector() = default;
  const std::string& getClassName() const { return rejection_name_; }
  virtual void getRemainingCorrespondences(const pcl::Correspondences&, pcl::Correspondences&) {
    mock_no_cpu_path("CorrespondenceRejector::getRemainingCorrespondences");
  }
 protected:
  std::string rejection_name_;
};
class CorrespondenceRejectorDistance : public CorrespondenceRejector {
 public:
  CorrespondenceRejectorDistance() { rejection_name_ = "CorrespondenceRejectorDistance"; }
  virtual void setMaximumDistance(float distance) { max_distance_ = distance * distance; }
  float getMaximumDistance() const { return std::sqrt(max_distance_); }
 protected:
  float max_distance_ = std::numeric_limits<float>::max();
};
class CorrespondenceRejectorMedianDistance : public CorrespondenceRejector {
 public:
  CorrespondenceRejectorMedianDistance() { rejection_name_ = "CorrespondenceRejectorMedianDistance"; }
  void setMedianFactor(double factor) { factor_ = factor; }
  double getMedianFactor() const { return factor_; }
 protected:
  double factor_ = 1.0;
};
class CorrespondenceRejectorOneToOne : public CorrespondenceRejector {
 public:
  CorrespondenceRejectorOneToOne() { rejection_name_ = "CorrespondenceRejectorOneToOne"; }
};
class CorrespondenceRejectorTrimmed : public CorrespondenceRejector {
 public:
  CorrespondenceRejectorTrimmed() { rejection_name_ = "CorrespondenceRejectorTrimmed"; }
  virtual void setOverlapRatio(float ratio) { overlap_ratio_ = ratio < 1.0f ? ratio : 1.0f; }
  float getOverlapRatio() const { return overlap_ratio_; }
  void setMinCorrespondences(unsigned int min_correspondences) { nr_min_correspondences_ = min_correspondences; }
  unsigned int getMinCorrespondences() const { return nr_min_correspondences_; }
 protected:
  float overlap_ratio_ = 0.5f;
  unsigned int nr_min_correspondences_ = 0;
};

// registration/include/pcl/registration/default_convergence_criteria.h:60-330 (state + thresholds only)
template <typename Scalar = float>
class DefaultConvergenceCriteria {
 public:
  using Ptr = shared_ptr<DefaultConvergenceCriteria<Scalar>>;
  using Matrix4 = Eigen::Matrix<Scalar, 4, 4>;
  enum ConvergenceState {
    CONVERGENCE_CRITERIA_NOT_CONVERGED, CONVERGENCE_CRITERIA_ITERATIONS, CONVERGENCE_CRITERIA_TRANSFORM,
    CONVERGENCE_CRITERIA_ABS_MSE, CONVERGENCE_CRITERIA_REL_MSE, CONVERGENCE_CRITERIA_NO_CORRESPONDENCES,
    CONVERGENCE_CRITERIA_FAILURE_AFTER_MAX_ITERATIONS
  };
  DefaultConvergenceCriteria(const int& iterations, const Matrix4& transform, const pcl::Correspondences& correspondences)
      : iterations_(iterations), transformation_(transform), correspondences_(correspondences) {}
  void setMaximumIterationsSimilarTransforms(int n) { max_iterations_similar_transforms_ = n; }
  int getMaximumIterationsSimilarTransforms() const { return max_iterations_similar_transforms_; }
  void setFailureAfterMaximumIterations(bool f) { failure_after_max_iter_ = f; }
  bool getFailureAfterMaximumIterations() const { return failure_after_max_iter_; }
  void setAbsoluteMSE(double mse) { mse_threshold_absolute_ = mse; }
  double getAbsoluteMSE() const { return mse_threshold_absolute_; }
  ConvergenceState getConvergenceState() { return convergence_state_; }
  void setConvergenceState(ConvergenceState c) { convergence_state_ = c; }
  bool hasConverged() { mock_no_cpu_path("DefaultConvergenceCriteria::hasConverged"); }
 protected:
  const int& iterations_;
  const Matrix4& transformation_;
  const pcl::Correspondences& correspondences_;
  double mse_threshold_absolute_ = 1e-12;
  int max_iterations_similar_transforms_ = 0;
  bool failure_after_max_iter_ = false;
  ConvergenceState convergence_state_ = CONVERGENCE_CRITERIA_NOT_CONVERGED;
};

// registration/include/pcl/registration/transformation_estimation.h:50-125
template <typename PointSource, typename PointTarget, typename Scalar = float>
class TransformationEstimation {
 public:
  using Matrix4 = Eigen::Matrix<Scalar, 4, 4>;
  using Ptr = shared_ptr<TransformationEstimation<PointSource, PointTarget, Scalar>>;
  using ConstPtr = shared_ptr<const TransformationEstimation<PointSource, PointTarget, Scalar>>;
  TransformationEstimation() = default;
  virtual ~TransformationEstimation() = default;
  virtual void estimateRigidTransformation(const pcl::PointCloud<PointSource>& cloud_src,
                                           const pcl::PointCloud<PointTarget>& cloud_tgt,
                                           Matrix4& transformation_matrix) const = 0;
  virtual void estimateRigidTransformation(const pcl::PointCloud<PointSource>& cloud_src, const pcl::Indices& indices_src,
                                           const pcl::PointCloud<PointTarget>& cloud_tgt,
                                           Matrix4& transformation_matrix) const = 0;
  virtual void estimateRigidTransformation(const pcl::PointCloud<PointSource>& cloud_src, const pcl::Indices& indices_src,
                                           const pcl::PointCloud<PointTarget>& cloud_tgt, const pcl::Indices& indices_tgt,
                                           Matrix4& transformation_matrix) const = 0;
  virtual void estimateRigidTransformation(const pcl::PointCloud<PointSource>& cloud_src,
                                           const pcl::PointCloud<PointTarget>& cloud_tgt,
                                           const pcl::Correspondences& correspondences,
                                           Matrix4& transformation_matrix) const = 0;
};
// the three estimators of the path (transformation_estimation_svd.h:56-140, ..._point_to_plane_lls.h:58-140,
// ..._symmetric_point_to_plane_lls.h:55-150): types only, their CPU bodies are not part of the mock
#define PCL_MOCK_ESTIMATOR(NAME)                                                                                          \
  template <typename PointSource, typename PointTarget, typename Scalar = float>                                          \
  class NAME : public TransformationEstimation<PointSource, PointTarget, Scalar> {                                       \
   public:                                                                                                                \
    using Matrix4 = typename TransformationEstimation<PointSource, PointTarget, Scalar>::Matrix4;                        \
    using Ptr = shared_ptr<NAME<PointSource, PointTarget, Scalar>>;                                                       \
    void estimateRigidTransformation(const pcl::PointCloud<PointSource>&, const pcl::PointCloud<PointTarget>&,            \
                                     Matrix4&) const override { mock_no_cpu_path(#NAME); }                                \
    void estimateRigidTransformation(const pcl::PointCloud<PointSource>&, const pcl::Indices&,                            \
                                     const pcl::PointCloud<PointTarget>&, Matrix4&) const override { mock_no_cpu_path(#NAME); } \
    void estimateRigidTransformation(const pcl::PointCloud<PointSource>&, const pcl::Indices&,                            \
                                     const pcl::PointCloud<PointTarget>&, const pcl::Indices&, Matrix4&) const override { \
      mock_no_cpu_path(#NAME);                                                                                            \
    }                                                                                                                     \
    void estimateRigidTransformation(const pcl::PointCloud<PointSource>&, const pcl::PointCloud<PointTarget>&,            \
                                     const pcl::Correspondences&, Matrix4&) const override { mock_no_cpu_path(#NAME); }   \
  }
PCL_MOCK_ESTIMATOR(TransformationEstimationSVD);
PCL_MOCK_ESTIMATOR(TransformationEstimationPointToPlaneLLS);
template <typename PointSource, typename PointTarget, typename Scalar = float>
class TransformationEstimationSymmetricPointToPlaneLLS : public TransformationEstimation<PointSource, PointTarget, Scalar> {
 public:
  using Matrix4 = typename TransformationEstimation<PointSource, PointTarget, Scalar>::Matrix4;
  using Ptr = shared_ptr<TransformationEstimationSymmetricPointToPlaneLLS<PointSource, PointTarget, Scalar>>;
  void estimateRigidTransformation(const pcl::PointCloud<PointSource>&, const pcl::PointCloud<PointTarget>&,
                                   Matrix4&) const override { mock_no_cpu_path("SymmetricPointToPlaneLLS"); }
  void estimateRigidTransformation(const pcl::PointCloud<PointSource>&, const pcl::Indices&,
                                   const pcl::PointCloud<PointTarget>&, Matrix4&) const override { mock_no_cpu_path("SymmetricPointToPlaneLLS"); }
  void estimateRigidTransformation(const pcl::PointCloud<PointSource>&, const pcl::Indices&,
                                   const pcl::PointCloud<PointTarget>&, const pcl::Indices&, Matrix4&) const override {
    mock_no_cpu_path("SymmetricPointToPlaneLLS");
  }
  void estimateRigidTransformation(const pcl::PointCloud<PointSource>&, const pcl::PointCloud<PointTarget>&,
                                   const pcl::Correspondences&, Matrix4&) const override { mock_no_cpu_path("SymmetricPointToPlaneLLS"); }
  void setEnforceSameDirectionNormals(bool e) { enforce_same_direction_normals_ = e; }
  bool getEnforceSameDirectionNormals() { return enforce_same_direction_normals_; }
 private:
  bool enforce_same_direction_normals_ = true;
};

// registration/include/pcl/registration/correspondence_estimation.h:62-330
template <typename PointSource, typename PointTarget, typename Scalar = float>
class CorrespondenceEstimationBase : public PCLBase<PointSource> {
 public:
  using Ptr = shared_ptr<CorrespondenceEstimationBase<PointSource, PointTarget, Scalar>>;
  using ConstPtr = shared_ptr<const CorrespondenceEstimationBase<PointSource, PointTarget, Scalar>>;
  using PCLBase<PointSource>::deinitCompute;
  using PCLBase<PointSource>::input_;
  using PCLBase<PointSource>::indices_;
  using PCLBase<PointSource>::setIndices;
  using KdTree = pcl::search::KdTree<PointTarget>;
  using KdTreePtr = typename KdTree::Ptr;
  using KdTreeReciprocal = pcl::search::KdTree<PointSource>;
  using KdTreeReciprocalPtr = typename KdTreeReciprocal::Ptr;
  using PointCloudSource = pcl::PointCloud<PointSource>;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = pcl::PointCloud<PointTarget>;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
  using PointRepresentationConstPtr = typename KdTree::PointRepresentationConstPtr;

  // the reference creates default trees here (:88-95); the mock has no CPU tree: a backend must be set
  CorrespondenceEstimationBase() : corr_name_("CorrespondenceEstimationBase") {}
  ~CorrespondenceEstimationBase() override = default;
  void setInputSource(const PointCloudSourceConstPtr& cloud) {
    source_cloud_updated_ = true;
    PCLBase<PointSource>::setInputCloud(cloud);
    input_fields_updated_ = true;
  }
  PointCloudSourceConstPtr const getInputSource() { return input_; }
  void setInputTarget(const PointCloudTargetConstPtr& cloud) {  // impl/correspondence_estimation.hpp:53-69
    if (cloud->points.empty()) return;
    target_ = cloud;
    if (point_representation_ && tree_) tree_->setPointRepresentation(point_representation_);
    target_cloud_updated_ = true;
  }
  PointCloudTargetConstPtr const getInputTarget() { return target_; }
  virtual bool requiresSourceNormals() const { return false; }
  virtual bool requiresTargetNormals() const { return false; }
  void setIndicesSource(const IndicesPtr& indices) { setIndices(indices); }
  IndicesPtr const getIndicesSource() { return indices_; }
  void setIndicesTarget(const IndicesPtr& indices) { target_cloud_updated_ = true; target_indices_ = indices; }
  IndicesPtr const getIndicesTarget() { return target_indices_; }
  void setSearchMethodTarget(const KdTreePtr& tree, bool force_no_recompute = false) {
    tree_ = tree;
    force_no_recompute_ = force_no_recompute;
    target_cloud_updated_ = true;
  }
  KdTreePtr getSearchMethodTarget() const { return tree_; }
  void setSearchMethodSource(const KdTreeReciprocalPtr& tree, bool force_no_recompute = false) {
    tree_reciprocal_ = tree;
    force_no_recompute_reciprocal_ = force_no_recompute;
    source_cloud_updated_ = true;
  }
  KdTreeReciprocalPtr getSearchMethodSource() const { return tree_reciprocal_; }
  virtual void determineCorrespondences(pcl::Correspondences& correspondences,
                                        double max_distance = std::numeric_limits<double>::max()) = 0;
  virtual void determineReciprocalCorrespondences(pcl::Correspondences& correspondences,
                                                  double max_distance = std::numeric_limits<double>::max()) = 0;
  void setPointRepresentation(const PointRepresentationConstPtr& point_representation) {
    point_representation_ = point_representation;
  }
  virtual Ptr clone() const = 0;
 protected:
  std::string corr_name_;
  KdTreePtr tree_;
  KdTreeReciprocalPtr tree_reciprocal_;
  PointCloudTargetConstPtr target_;
  IndicesPtr target_indices_;
  PointRepresentationConstPtr point_representation_;
  PointCloudTargetConstPtr input_transformed_;
  std::vector<int> input_fields_;
  const std::string& getClassName() const { return corr_name_; }
  bool initCompute() {  // impl/correspondence_estimation.hpp:71-97
    if (!target_ || !tree_) return false;
    if (target_cloud_updated_ && !force_no_recompute_) {
      if (target_indices_) tree_->setInputCloud(target_, target_indices_);
      else tree_->setInputCloud(target_);
      target_cloud_updated_ = false;
    }
    return PCLBase<PointSource>::initCompute();
  }
  bool initComputeReciprocal() {  // :99-113
    if (tree_reciprocal_ && source_cloud_updated_ && !force_no_recompute_reciprocal_) {
      if (point_representation_) tree_reciprocal_->setPointRepresentation(point_representation_);
      tree_reciprocal_->setInputCloud(PCLBase<PointSource>::getInputCloud(), PCLBase<PointSource>::getIndices());
      source_cloud_updated_ = false;
    }
    return true;
  }
  bool target_cloud_updated_ = true, source_cloud_updated_ = true;
  bool force_no_recompute_ = false, force_no_recompute_reciprocal_ = false;
  bool input_fields_updated_ = false;
};

// PCL's own estimator: ONE nearestKSearch per source point through the virtual search interface, from an OpenMP loop
// with setNumberOfThreads(n) threads (correspondence_estimation.h:144-155, impl/correspondence_estimation.hpp:145-218)
// -- the call pattern a search backend alone is subject to, concurrency included
template <typename PointSource, typename PointTarget, typename Scalar = float>
class CorrespondenceEstimation : public CorrespondenceEstimationBase<PointSource, PointTarget, Scalar> {
  using Base = CorrespondenceEstimationBase<PointSource, PointTarget, Scalar>;
 public:
  using Ptr = shared_ptr<CorrespondenceEstimation<PointSource, PointTarget, Scalar>>;
  CorrespondenceEstimation() { this->corr_name_ = "CorrespondenceEstimation"; }
  void setNumberOfThreads(unsigned int nr_threads) {  // correspondence_estimation.h:144-155
#ifdef _OPENMP
    num_threads_ = nr_threads != 0 ? nr_threads : static_cast<unsigned int>(omp_get_num_procs());
#else
    (void)nr_threads;
    num_threads_ = 1;
#endif
  }
  void determineCorrespondences(pcl::Correspondences& correspondences,
                                double max_distance = std::numeric_limits<double>::max()) override {
    correspondences.clear();
    if (!this->initCompute()) return;
    const double max_dist_sqr = max_distance * max_distance;
    std::vector<pcl::Correspondences> per_thread(num_threads_);
    const auto& indices = *this->indices_;
    const auto& input = *this->input_;
    const auto& tree = *this->tree_;
    const int n = static_cast<int>(indices.size());
#ifdef _OPENMP
#pragma omp parallel for num_threads(num_threads_)
#endif
    for (int i = 0; i < n; ++i) {
      pcl::Indices index(1);            // firstprivate(index, distance) in the reference
      std::vector<float> distance(1);
      const index_t idx = indices[std::size_t(i)];
      const PointSource& p = input[idx];
      if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
      if (tree.nearestKSearchT(p, 1, index, distance) == 0) continue;
      if (distance[0] > max_dist_sqr) continue;
#ifdef _OPENMP
      const int t = omp_get_thread_num();
#else
      const int t = 0;
#endif
      per_thread[std::size_t(t)].emplace_back(idx, index[0], distance[0]);
    }
    // :191-214: the per-thread lists merged, ordered by index_query
    for (const auto& c : per_thread) correspondences.insert(correspondences.end(), c.begin(), c.end());
    if (num_threads_ > 1)
      std::stable_sort(correspondences.begin(), correspondences.end(),
                       [](const pcl::Correspondence& a, const pcl::Correspondence& b) { return a.index_query < b.index_query; });
    this->deinitCompute();
  }
  void determineReciprocalCorrespondences(pcl::Correspondences&, double = std::numeric_limits<double>::max()) override {
    mock_no_cpu_path("CorrespondenceEstimation::determineReciprocalCorrespondences");
  }
  typename Base::Ptr clone() const override {
    return typename Base::Ptr(new CorrespondenceEstimation<PointSource, PointTarget, Scalar>(*this));
  }
 protected:
  unsigned int num_threads_{1};  // correspondence_estimation.h:383
};

}  // namespace registration

// registration/include/pcl/registration/registration.h:56-700
template <typename PointSource, typename PointTarget, typename Scalar = float>
class Registration : public PCLBase<PointSource> {
 public:
  using Matrix4 = Eigen::Matrix<Scalar, 4, 4>;
  using PCLBase<PointSource>::deinitCompute;
  using PCLBase<PointSource>::input_;
  using PCLBase<PointSource>::indices_;
  using Ptr = shared_ptr<Registration<PointSource, PointTarget, Scalar>>;
  using CorrespondenceRejectorPtr = pcl::registration::CorrespondenceRejector::Ptr;
  using KdTree = pcl::search::KdTree<PointTarget>;
  using KdTreePtr = typename KdTree::Ptr;
  using KdTreeReciprocal = pcl::search::KdTree<PointSource>;
  using KdTreeReciprocalPtr = typename KdTreeReciprocal::Ptr;
  using PointCloudSource = pcl::PointCloud<PointSource>;
  using PointCloudSourcePtr = typename PointCloudSource::Ptr;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = pcl::PointCloud<PointTarget>;
  using PointCloudTargetPtr = typename PointCloudTarget::Ptr;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
  using PointRepresentationConstPtr = typename KdTree::PointRepresentationConstPtr;
  using TransformationEstimation = typename pcl::registration::TransformationEstimation<PointSource, PointTarget, Scalar>;
  using TransformationEstimationPtr = typename TransformationEstimation::Ptr;
  using CorrespondenceEstimation = pcl::registration::CorrespondenceEstimationBase<PointSource, PointTarget, Scalar>;
  using CorrespondenceEstimationPtr = typename CorrespondenceEstimation::Ptr;

  // the reference constructs default FLANN trees here (:113-118); the mock has none
  Registration()
      : final_transformation_(Matrix4::Identity()), transformation_(Matrix4::Identity()),
        previous_transformation_(Matrix4::Identity()), euclidean_fitness_epsilon_(-std::numeric_limits<double>::max()),
        corr_dist_threshold_(std::sqrt(std::numeric_limits<double>::max())), correspondences_(new Correspondences) {}
  ~Registration() override = default;
  void setTransformationEstimation(const TransformationEstimationPtr& te) { transformation_estimation_ = te; }
  void setCorrespondenceEstimation(const CorrespondenceEstimationPtr& ce) { correspondence_estimation_ = ce; }
  virtual void setInputSource(const PointCloudSourceConstPtr& cloud) {  // impl/registration.hpp:45-56
    if (cloud->points.empty()) return;
    source_cloud_updated_ = true;
    PCLBase<PointSource>::setInputCloud(cloud);
  }
  PointCloudSourceConstPtr const getInputSource() { return input_; }
  virtual void setInputTarget(const PointCloudTargetConstPtr& cloud) {  // :58-69
    if (cloud->points.empty()) return;
    target_ = cloud;
    target_cloud_updated_ = true;
  }
  PointCloudTargetConstPtr const getInputTarget() { return target_; }
  void setSearchMethodTarget(const KdTreePtr& tree, bool force_no_recompute = false) {
    tree_ = tree;
    force_no_recompute_ = force_no_recompute;
    target_cloud_updated_ = true;
  }
  KdTreePtr getSearchMethodTarget() const { return tree_; }
  void setSearchMethodSource(const KdTreeReciprocalPtr& tree, bool force_no_recompute = false) {
    tree_reciprocal_ = tree;
    force_no_recompute_reciprocal_ = force_no_recompute;
    source_cloud_updated_ = true;
  }
  KdTreeReciprocalPtr getSearchMethodSource() const { return tree_reciprocal_; }
  Matrix4 getFinalTransformation() { return final_transformation_; }
  Matrix4 getLastIncrementalTransformation() { return transformation_; }
  void setMaximumIterations(int nr_iterations) { max_iterations_ = nr_iterations; }
  int getMaximumIterations() { return max_iterations_; }
  void setMaxCorrespondenceDistance(double distance_threshold) { corr_dist_threshold_ = distance_threshold; }
  double getMaxCorrespondenceDistance() { return corr_dist_threshold_; }
  void setTransformationEpsilon(double epsilon) { transformation_epsilon_ = epsilon; }
  double getTransformationEpsilon() { return transformation_epsilon_; }
  void setTransformationRotationEpsilon(double epsilon) { transformation_rotation_epsilon_ = epsilon; }
  double getTransformationRotationEpsilon() { return transformation_rotation_epsilon_; }
  void setEuclideanFitnessEpsilon(double epsilon) { euclidean_fitness_epsilon_ = epsilon; }
  double getEuclideanFitnessEpsilon() { return euclidean_fitness_epsilon_; }
  void setPointRepresentation(const PointRepresentationConstPtr& point_representation) {
    point_representation_ = point_representation;
  }
  // registration.h:450-452 -- NOT virtual in the reference: a subclass can only hide it.  impl/registration.hpp:132-168:
  // the moved source through tree_->nearestKSearchT point by point (the one CPU loop the mock carries: it is how a
  // caller holding a pcl::Registration* reaches the search backend)
  inline double getFitnessScore(double max_range = std::numeric_limits<double>::max(), bool use_indices = false) {
    double fitness_score = 0.0;
    const bool subset = use_indices && indices_ && indices_->size() != input_->size();
    const std::size_t n = subset ? indices_->size() : input_->size();
    pcl::Indices nn_indices(1);
    std::vector<float> nn_dists(1);
    int nr = 0;
    const Matrix4& T = final_transformation_;
    for (std::size_t i = 0; i < n; ++i) {
      const PointSource& s = (*input_)[subset ? std::size_t((*indices_)[i]) : i];
      PointSource p = s;  // transformPointCloud: float, rotation rows then the translation (common/.../impl/transforms.hpp)
      p.x = float(T(0, 0) * s.x + T(0, 1) * s.y + T(0, 2) * s.z + T(0, 3));
      p.y = float(T(1, 0) * s.x + T(1, 1) * s.y + T(1, 2) * s.z + T(1, 3));
      p.z = float(T(2, 0) * s.x + T(2, 1) * s.y + T(2, 2) * s.z + T(2, 3));
      if (!input_->is_dense && !(std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z))) continue;
      tree_->nearestKSearchT(p, 1, nn_indices, nn_dists);
      if (nn_dists[0] <= max_range) {
        fitness_score += nn_dists[0];
        nr++;
      }
    }
    if (nr > 0) return fitness_score / nr;
    return std::numeric_limits<double>::max();
  }
  bool hasConverged() const { return converged_; }
  void align(PointCloudSource& output) { align(output, Matrix4::Identity()); }
  void align(PointCloudSource& output, const Matrix4& guess) {  // impl/registration.hpp:178-221
    if (!initCompute()) return;
    output.resize(indices_->size());
    output.header = input_->header;
    if (indices_->size() != input_->size()) { output.width = std::uint32_t(indices_->size()); output.height = 1; }
    else { output.width = input_->width; output.height = input_->height; }
    output.is_dense = input_->is_dense;
    for (std::size_t i = 0; i < indices_->size(); ++i) output[i] = (*input_)[(*indices_)[i]];
    if (point_representation_ && !force_no_recompute_) tree_->setPointRepresentation(point_representation_);
    converged_ = false;
    final_transformation_ = transformation_ = previous_transformation_ = Matrix4::Identity();
    for (std::size_t i = 0; i < indices_->size(); ++i) output[i].data[3] = 1.0;
    computeTransformation(output, guess);
    deinitCompute();
  }
  const std::string& getClassName() const { return reg_name_; }
  bool initCompute() {  // impl/registration.hpp:73-101
    if (!target_ || !tree_) return false;
    if (target_cloud_updated_ && !force_no_recompute_) {
      tree_->setInputCloud(target_);
      target_cloud_updated_ = false;
    }
    if (correspondence_estimation_) {
      correspondence_estimation_->setSearchMethodTarget(tree_, force_no_recompute_);
      correspondence_estimation_->setSearchMethodSource(tree_reciprocal_, force_no_recompute_reciprocal_);
    }
    return PCLBase<PointSource>::initCompute();
  }
  void addCorrespondenceRejector(const CorrespondenceRejectorPtr& rejector) { correspondence_rejectors_.push_back(rejector); }
  std::vector<CorrespondenceRejectorPtr> getCorrespondenceRejectors() { return correspondence_rejectors_; }
  bool removeCorrespondenceRejector(unsigned int i) {
    if (i >= correspondence_rejectors_.size()) return false;
    correspondence_rejectors_.erase(correspondence_rejectors_.begin() + i);
    return true;
  }
  void clearCorrespondenceRejectors() { correspondence_rejectors_.clear(); }
 protected:
  std::string reg_name_;
  KdTreePtr tree_;
  KdTreeReciprocalPtr tree_reciprocal_;
  int nr_iterations_{0};
  int max_iterations_{10};
  int ransac_iterations_{0};
  PointCloudTargetConstPtr target_;
  Matrix4 final_transformation_, transformation_, previous_transformation_;
  double transformation_epsilon_{0.0};
  double transformation_rotation_epsilon_{0.0};
  double euclidean_fitness_epsilon_;
  double corr_dist_threshold_;
  double inlier_threshold_{0.05};
  bool converged_{false};
  unsigned int min_number_correspondences_{3};
  CorrespondencesPtr correspondences_;
  TransformationEstimationPtr transformation_estimation_;
  CorrespondenceEstimationPtr correspondence_estimation_;
  std::vector<CorrespondenceRejectorPtr> correspondence_rejectors_;
  bool target_cloud_updated_{true};
  bool source_cloud_updated_{true};
  bool force_no_recompute_{false};
  bool force_no_recompute_reciprocal_{false};
  virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) = 0;  // :678-679
 private:
  PointRepresentationConstPtr point_representation_;
  void setInputCloud(const PointCloudSourceConstPtr& cloud) override { setInputSource(cloud); }
};

// registration/include/pcl/registration/icp.h:98-347
template <typename PointSource, typename PointTarget, typename Scalar = float>
class IterativeClosestPoint : public Registration<PointSource, PointTarget, Scalar> {
 public:
  using PointCloudSource = typename Registration<PointSource, PointTarget, Scalar>::PointCloudSource;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = typename Registration<PointSource, PointTarget, Scalar>::PointCloudTarget;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
  using Ptr = shared_ptr<IterativeClosestPoint<PointSource, PointTarget, Scalar>>;
  using Matrix4 = typename Registration<PointSource, PointTarget, Scalar>::Matrix4;
  using Registration<PointSource, PointTarget, Scalar>::reg_name_;
  using Registration<PointSource, PointTarget, Scalar>::nr_iterations_;
  using Registration<PointSource, PointTarget, Scalar>::transformation_;
  using Registration<PointSource, PointTarget, Scalar>::correspondences_;
  using Registration<PointSource, PointTarget, Scalar>::transformation_estimation_;
  using Registration<PointSource, PointTarget, Scalar>::correspondence_estimation_;
  typename pcl::registration::DefaultConvergenceCriteria<Scalar>::Ptr convergence_criteria_;

  IterativeClosestPoint() {  // :136-151
    reg_name_ = "IterativeClosestPoint";
    transformation_estimation_.reset(new pcl::registration::TransformationEstimationSVD<PointSource, PointTarget, Scalar>());
    correspondence_estimation_.reset(new pcl::registration::CorrespondenceEstimation<PointSource, PointTarget, Scalar>);
    convergence_criteria_.reset(
        new pcl::registration::DefaultConvergenceCriteria<Scalar>(nr_iterations_, transformation_, *correspondences_));
  }
  IterativeClosestPoint(const IterativeClosestPoint&) = delete;
  IterativeClosestPoint& operator=(const IterativeClosestPoint&) = delete;
  ~IterativeClosestPoint() override = default;
  typename pcl::registration::DefaultConvergenceCriteria<Scalar>::Ptr getConvergeCriteria() { return convergence_criteria_; }
  void setUseReciprocalCorrespondences(bool use_reciprocal_correspondence) {
    use_reciprocal_correspondence_ = use_reciprocal_correspondence;
  }
  bool getUseReciprocalCorrespondences() const { return use_reciprocal_correspondence_; }
 protected:
  virtual void transformCloud(const PointCloudSource&, PointCloudSource&, const Matrix4&) {
    mock_no_cpu_path("IterativeClosestPoint::transformCloud");
  }
  void computeTransformation(PointCloudSource&, const Matrix4&) override {  // :292-293
    mock_no_cpu_path("IterativeClosestPoint::computeTransformation");
  }
  bool use_reciprocal_correspondence_{false};
  bool source_has_normals_{false};
  bool target_has_normals_{false};
};

// registration/include/pcl/registration/icp.h:360-440
template <typename PointSource, typename PointTarget, typename Scalar = float>
class IterativeClosestPointWithNormals : public IterativeClosestPoint<PointSource, PointTarget, Scalar> {
 public:
  using PointCloudSource = typename IterativeClosestPoint<PointSource, PointTarget, Scalar>::PointCloudSource;
  using Matrix4 = typename IterativeClosestPoint<PointSource, PointTarget, Scalar>::Matrix4;
  using IterativeClosestPoint<PointSource, PointTarget, Scalar>::reg_name_;
  using IterativeClosestPoint<PointSource, PointTarget, Scalar>::transformation_estimation_;
  using Ptr = shared_ptr<IterativeClosestPoint<PointSource, PointTarget, Scalar>>;
  IterativeClosestPointWithNormals() {
    reg_name_ = "IterativeClosestPointWithNormals";
    setUseSymmetricObjective(false);
    setEnforceSameDirectionNormals(true);
  }
  void setUseSymmetricObjective(bool use_symmetric_objective) {  // :380-400
    use_symmetric_objective_ = use_symmetric_objective;
    if (use_symmetric_objective_)
      transformation_estimation_.reset(
          new pcl::registration::TransformationEstimationSymmetricPointToPlaneLLS<PointSource, PointTarget, Scalar>());
    else
      transformation_estimation_.reset(
          new pcl::registration::TransformationEstimationPointToPlaneLLS<PointSource, PointTarget, Scalar>());
  }
  bool getUseSymmetricObjective() const { return use_symmetric_objective_; }
  void setEnforceSameDirectionNormals(bool enforce_same_direction_normals) {
    enforce_same_direction_normals_ = enforce_same_direction_normals;
  }
  bool getEnforceSameDirectionNormals() const { return enforce_same_direction_normals_; }
 protected:
  bool use_symmetric_objective_ = false;
  bool enforce_same_direction_normals_ = true;
};


// common/include/pcl/PCLPointField.h:12-35, common/include/pcl/common/io.h:78-96 (the field list of a point type)
struct PCLPointField {
  std::string name;
  std::uint32_t offset = 0;
  std::uint8_t datatype = 0;
  std::uint32_t count = 0;
  enum PointFieldTypes { FLOAT32 = 7 };
};
template <typename PointT> inline std::vector<PCLPointField> getFields();
template <> inline std::vector<PCLPointField> getFields<PointXYZ>() {
  return {{"x", 0, PCLPointField::FLOAT32, 1}, {"y", 4, PCLPointField::FLOAT32, 1}, {"z", 8, PCLPointField::FLOAT32, 1}};
}
template <> inline std::vector<PCLPointField> getFields<PointNormal>() {  // impl/point_types.hpp:843-853
  return {{"x", 0, PCLPointField::FLOAT32, 1},         {"y", 4, PCLPointField::FLOAT32, 1},
          {"z", 8, PCLPointField::FLOAT32, 1},         {"normal_x", 16, PCLPointField::FLOAT32, 1},
          {"normal_y", 20, PCLPointField::FLOAT32, 1}, {"normal_z", 24, PCLPointField::FLOAT32, 1},
          {"curvature", 32, PCLPointField::FLOAT32, 1}};
}
template <> inline std::vector<PCLPointField> getFields<PointXYZI>() {
  return {{"x", 0, PCLPointField::FLOAT32, 1}, {"y", 4, PCLPointField::FLOAT32, 1}, {"z", 8, PCLPointField::FLOAT32, 1},
          {"intensity", 16, PCLPointField::FLOAT32, 1}};
}
template <> inline std::vector<PCLPointField> getFields<PointXYZRGBNormal>() {
  return {{"x", 0, PCLPointField::FLOAT32, 1},         {"y", 4, PCLPointField::FLOAT32, 1},
          {"z", 8, PCLPointField::FLOAT32, 1},         {"normal_x", 16, PCLPointField::FLOAT32, 1},
          {"normal_y", 20, PCLPointField::FLOAT32, 1}, {"normal_z", 24, PCLPointField::FLOAT32, 1},
          {"rgb", 32, PCLPointField::FLOAT32, 1},      {"curvature", 36, PCLPointField::FLOAT32, 1}};
}
template <typename PointT> inline int getFieldIndex(const std::string& field_name, std::vector<PCLPointField>& fields) {
  fields = getFields<PointT>();
  for (std::size_t i = 0; i < fields.size(); ++i)
    if (fields[i].name == field_name) return int(i);
  return -1;
}
struct PointIndices { PCLHeader header; Indices indices; };
template <typename PointT> inline void copyPointCloud(const PointCloud<PointT>& in, PointCloud<PointT>& out) { out = in; }

// features/include/pcl/features/feature.h:100-316, impl/feature.hpp:95-229
template <typename PointInT, typename PointOutT>
class Feature : public PCLBase<PointInT> {
 public:
  using PCLBase<PointInT>::indices_;
  using PCLBase<PointInT>::input_;
  using BaseClass = PCLBase<PointInT>;
  using Ptr = shared_ptr<Feature<PointInT, PointOutT>>;
  using ConstPtr = shared_ptr<const Feature<PointInT, PointOutT>>;
  using KdTree = pcl::search::Search<PointInT>;
  using KdTreePtr = typename KdTree::Ptr;
  using PointCloudIn = pcl::PointCloud<PointInT>;
  using PointCloudInPtr = typename PointCloudIn::Ptr;
  using PointCloudInConstPtr = typename PointCloudIn::ConstPtr;
  using PointCloudOut = pcl::PointCloud<PointOutT>;
  using SearchMethodSurface = std::function<int(const PointCloudIn& cloud, std::size_t index, double, pcl::Indices&, std::vector<float>&)>;

  Feature() : search_parameter_(0), search_radius_(0), k_(0), fake_surface_(false) {}
  inline void setSearchSurface(const PointCloudInConstPtr& cloud) { surface_ = cloud; fake_surface_ = false; }
  inline PointCloudInConstPtr getSearchSurface() const { return surface_; }
  inline void setSearchMethod(const KdTreePtr& tree) { tree_ = tree; }
  inline KdTreePtr getSearchMethod() const { return tree_; }
  inline double getSearchParameter() const { return search_parameter_; }
  inline void setKSearch(int k) { k_ = k; }
  inline int getKSearch() const { return k_; }
  inline void setRadiusSearch(double radius) { search_radius_ = radius; }
  inline double getRadiusSearch() const { return search_radius_; }
  void compute(PointCloudOut& output) {  // impl/feature.hpp:195-229
    if (!initCompute()) {
      output.width = output.height = 0;
      output.points.clear();
      return;
    }
    output.header = input_->header;
    if (output.size() != indices_->size()) output.resize(indices_->size());
    if (indices_->size() != input_->points.size() || input_->width * input_->height == 0) {
      output.width = std::uint32_t(indices_->size());
      output.height = 1;
    } else {
      output.width = input_->width;
      output.height = input_->height;
    }
    output.is_dense = input_->is_dense;
    computeFeature(output);
    deinitCompute();
  }
 protected:
  std::string feature_name_;
  SearchMethodSurface search_method_surface_;
  PointCloudInConstPtr surface_;
  KdTreePtr tree_;
  double search_parameter_;
  double search_radius_;
  int k_;
  inline const std::string& getClassName() const { return feature_name_; }
  virtual bool initCompute() {  // impl/feature.hpp:95-178
    if (!PCLBase<PointInT>::initCompute()) return false;
    if (input_->points.empty()) { deinitCompute(); return false; }
    if (!surface_) { fake_surface_ = true; surface_ = input_; }
    if (!tree_) mock_no_cpu_path("pcl::search::autoSelectMethod");
    if (tree_->getInputCloud() != surface_) {
      if (!tree_->setInputCloud(surface_)) return false;
    }
    if (search_radius_ != 0.0) {
      if (k_ != 0) { deinitCompute(); return false; }  // "Both radius and K defined!"
      search_parameter_ = search_radius_;
      search_method_surface_ = [this](const PointCloudIn& cloud, std::size_t index, double radius, pcl::Indices& k_indices,
                                      std::vector<float>& k_distances) {
        return tree_->radiusSearch(cloud, index_t(index), radius, k_indices, k_distances, 0);
      };
    } else {
      if (k_ == 0) { deinitCompute(); return false; }  // "Neither radius nor K defined!"
      search_parameter_ = k_;
      search_method_surface_ = [this](const PointCloudIn& cloud, std::size_t index, double k, pcl::Indices& k_indices,
                                      std::vector<float>& k_distances) {
        return tree_->nearestKSearch(cloud, index_t(index), int(k), k_indices, k_distances);
      };
    }
    return true;
  }
  virtual bool deinitCompute() {
    if (fake_surface_) { surface_.reset(); fake_surface_ = false; }
    return true;
  }
  bool fake_surface_;
  inline int searchForNeighbors(std::size_t index, double parameter, pcl::Indices& indices, std::vector<float>& distances) const {
    return search_method_surface_(*input_, index, parameter, indices, distances);
  }
 private:
  virtual void computeFeature(PointCloudOut& output) = 0;
};

// features/include/pcl/features/normal_3d.h:236-410
template <typename PointInT, typename PointOutT>
class NormalEstimation : public Feature<PointInT, PointOutT> {
 public:
  using Ptr = shared_ptr<NormalEstimation<PointInT, PointOutT>>;
  using ConstPtr = shared_ptr<const NormalEstimation<PointInT, PointOutT>>;
  using Feature<PointInT, PointOutT>::feature_name_;
  using Feature<PointInT, PointOutT>::getClassName;
  using Feature<PointInT, PointOutT>::indices_;
  using Feature<PointInT, PointOutT>::input_;
  using Feature<PointInT, PointOutT>::surface_;
  using Feature<PointInT, PointOutT>::k_;
  using Feature<PointInT, PointOutT>::search_radius_;
  using Feature<PointInT, PointOutT>::search_parameter_;
  using PointCloudOut = typename Feature<PointInT, PointOutT>::PointCloudOut;
  using PointCloudConstPtr = typename Feature<PointInT, PointOutT>::PointCloudInConstPtr;

  NormalEstimation() { feature_name_ = "NormalEstimation"; }
  ~NormalEstimation() override = default;
  inline void setInputCloud(const PointCloudConstPtr& cloud) override {
    input_ = cloud;
    if (use_sensor_origin_) {
      vpx_ = input_->sensor_origin_.coeff(0);
      vpy_ = input_->sensor_origin_.coeff(1);
      vpz_ = input_->sensor_origin_.coeff(2);
    }
  }
  inline void setViewPoint(float vpx, float vpy, float vpz) { vpx_ = vpx; vpy_ = vpy; vpz_ = vpz; use_sensor_origin_ = false; }
  inline void getViewPoint(float& vpx, float& vpy, float& vpz) { vpx = vpx_; vpy = vpy_; vpz = vpz_; }
  inline void useSensorOriginAsViewPoint() {
    use_sensor_origin_ = true;
    if (input_) {
      vpx_ = input_->sensor_origin_.coeff(0); vpy_ = input_->sensor_origin_.coeff(1); vpz_ = input_->sensor_origin_.coeff(2);
    } else {
      vpx_ = 0; vpy_ = 0; vpz_ = 0;
    }
  }
 protected:
  void computeFeature(PointCloudOut&) override { mock_no_cpu_path("pcl::NormalEstimation::computeFeature"); }
  float vpx_{0.0f}, vpy_{0.0f}, vpz_{0.0f};
  bool use_sensor_origin_{true};
};

// filters/include/pcl/filters/filter.h:78-206
template <typename PointT>
class Filter : public PCLBase<PointT> {
 public:
  using Ptr = shared_ptr<Filter<PointT>>;
  using ConstPtr = shared_ptr<const Filter<PointT>>;
  using PointCloud = pcl::PointCloud<PointT>;
  using PointCloudPtr = typename PointCloud::Ptr;
  using PointCloudConstPtr = typename PointCloud::ConstPtr;
  Filter(bool extract_removed_indices = false) : removed_indices_(new Indices), extract_removed_indices_(extract_removed_indices) {}
  inline IndicesConstPtr const getRemovedIndices() const { return removed_indices_; }
  inline void getRemovedIndices(PointIndices& pi) { pi.indices = *removed_indices_; }
  inline void filter(PointCloud& output) {
    if (!initCompute()) return;
    if (input_.get() == &output) {  // cloud_in = cloud_out
      PointCloud output_temp;
      applyFilter(output_temp);
      output_temp.header = input_->header;
      output_temp.sensor_origin_ = input_->sensor_origin_;
      output_temp.sensor_orientation_ = input_->sensor_orientation_;
      pcl::copyPointCloud(output_temp, output);
    } else {
      output.header = input_->header;
      output.sensor_origin_ = input_->sensor_origin_;
      output.sensor_orientation_ = input_->sensor_orientation_;
      applyFilter(output);
    }
    deinitCompute();
  }
 protected:
  using PCLBase<PointT>::indices_;
  using PCLBase<PointT>::input_;
  using PCLBase<PointT>::initCompute;
  using PCLBase<PointT>::deinitCompute;
  IndicesPtr removed_indices_;
  std::string filter_name_;
  bool extract_removed_indices_;
  virtual void applyFilter(PointCloud& output) = 0;
  inline const std::string& getClassName() const { return filter_name_; }
};

// filters/include/pcl/filters/voxel_grid.h:210-533 (members and the setters / getters the binding relies on)
template <typename PointT>
class VoxelGrid : public Filter<PointT> {
 protected:
  using Filter<PointT>::filter_name_;
  using Filter<PointT>::getClassName;
  using Filter<PointT>::input_;
  using Filter<PointT>::indices_;
  using PointCloud = typename Filter<PointT>::PointCloud;
 public:
  using Ptr = shared_ptr<VoxelGrid<PointT>>;
  using ConstPtr = shared_ptr<const VoxelGrid<PointT>>;
  VoxelGrid() { filter_name_ = "VoxelGrid"; }
  ~VoxelGrid() override = default;
  inline void setLeafSize(const Eigen::Vector4f& leaf_size) {
    leaf_size_ = leaf_size;
    if (leaf_size_[3] == 0) leaf_size_[3] = 1;
    for (int d = 0; d < 4; ++d) inverse_leaf_size_[d] = 1.0f / leaf_size_[d];
  }
  inline void setLeafSize(float lx, float ly, float lz) {
    leaf_size_[0] = lx; leaf_size_[1] = ly; leaf_size_[2] = lz;
    if (leaf_size_[3] == 0) leaf_size_[3] = 1;
    for (int d = 0; d < 4; ++d) inverse_leaf_size_[d] = 1.0f / leaf_size_[d];
  }
  inline Eigen::Vector3f getLeafSize() const { Eigen::Vector3f v; v[0] = leaf_size_[0]; v[1] = leaf_size_[1]; v[2] = leaf_size_[2]; return v; }
  inline void setDownsampleAllData(bool downsample) { downsample_all_data_ = downsample; }
  inline bool getDownsampleAllData() const { return downsample_all_data_; }
  inline void setMinimumPointsNumberPerVoxel(unsigned int min_points_per_voxel) { min_points_per_voxel_ = min_points_per_voxel; }
  inline unsigned int getMinimumPointsNumberPerVoxel() const { return min_points_per_voxel_; }
  inline void setSaveLeafLayout(bool save_leaf_layout) { save_leaf_layout_ = save_leaf_layout; }
  inline bool getSaveLeafLayout() const { return save_leaf_layout_; }
  inline Eigen::Vector3i getMinBoxCoordinates() const { return head3(min_b_); }
  inline Eigen::Vector3i getMaxBoxCoordinates() const { return head3(max_b_); }
  inline Eigen::Vector3i getNrDivisions() const { return head3(div_b_); }
  inline Eigen::Vector3i getDivisionMultiplier() const { return head3(divb_mul_); }
  inline int getCentroidIndex(const PointT& p) const {  // voxel_grid.h:349-355
    return leaf_layout_.at(std::size_t((int(std::floor(p.x * inverse_leaf_size_[0])) - min_b_[0]) * divb_mul_[0] +
                                       (int(std::floor(p.y * inverse_leaf_size_[1])) - min_b_[1]) * divb_mul_[1] +
                                       (int(std::floor(p.z * inverse_leaf_size_[2])) - min_b_[2]) * divb_mul_[2]));
  }
  inline std::vector<int> getLeafLayout() const { return leaf_layout_; }
  inline void setFilterFieldName(const std::string& field_name) { filter_field_name_ = field_name; }
  inline std::string const getFilterFieldName() const { return filter_field_name_; }
  inline void setFilterLimits(const double& limit_min, const double& limit_max) { filter_limit_min_ = limit_min; filter_limit_max_ = limit_max; }
  inline void getFilterLimits(double& limit_min, double& limit_max) const { limit_min = filter_limit_min_; limit_max = filter_limit_max_; }
  inline void setFilterLimitsNegative(const bool limit_negative) { filter_limit_negative_ = limit_negative; }
  inline bool getFilterLimitsNegative() const { return filter_limit_negative_; }
 protected:
  static Eigen::Vector3i head3(const Eigen::Vector4i& v) { Eigen::Vector3i r; r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; return r; }
  Eigen::Vector4f leaf_size_;
  Eigen::Array4f inverse_leaf_size_;
  bool downsample_all_data_{true};
  bool save_leaf_layout_{false};
  std::vector<int> leaf_layout_;
  Eigen::Vector4i min_b_, max_b_, div_b_, divb_mul_;
  std::string filter_field_name_;
  double filter_limit_min_{std::numeric_limits<float>::lowest()};
  double filter_limit_max_{std::numeric_limits<float>::max()};
  bool filter_limit_negative_{false};
  unsigned int min_points_per_voxel_{0};
  void applyFilter(PointCloud&) override { mock_no_cpu_path("pcl::VoxelGrid::applyFilter"); }
};

}  // namespace pcl
