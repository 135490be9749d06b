// pcl_compat.hpp -- header-only C++ mirror of the PCL plugin surface for the ICP hot path, written against the
// C ABI of pclhip.h (no Eigen, no Boost, no FLANN) for applications that do not have PCL at all.  Same names,
// the same VIRTUAL structure, argument meaning and error behaviour as the reference classes, so user code
// (and PCL's own tests) port by changing the namespace -- including code that plugs its own search method,
// correspondence estimation or transformation estimation into a Registration:
//
//   pclhip::PointXYZ / PointNormal / Normal         common/include/pcl/impl/point_types.hpp:315-321,843-853,787-794
//   pclhip::PointCloud<PointT>                      common/include/pcl/point_cloud.h:173,393-409
//   pclhip::PCLBase<PointT>                         common/include/pcl/pcl_base.h:65-175 (setIndices)
//   pclhip::PointRepresentation<PointT> (+Default, Custom)   common/include/pcl/point_representation.h
//   pclhip::Correspondence(s)                       common/include/pcl/correspondence.h:60-91
//   pclhip::search::Search<PointT> (abstract)       search/include/pcl/search/search.h:60-420
//   pclhip::search::KdTree<PointT>                  search/include/pcl/search/kdtree.h:61-168
//   pclhip::registration::CorrespondenceEstimationBase (abstract) / CorrespondenceEstimation
//                                                   registration/include/pcl/registration/correspondence_estimation.h
//   pclhip::registration::TransformationEstimation (abstract, four overloads) / ...SVD / ...PointToPlaneLLS /
//   ...SymmetricPointToPlaneLLS                     registration/include/pcl/registration/transformation_estimation*.h
//   pclhip::registration::CorrespondenceRejector{Distance,MedianDistance,OneToOne,Trimmed}
//   pclhip::Registration<S,T> (abstract) / IterativeClosestPoint / IterativeClosestPointWithNormals
//                                                   registration/include/pcl/registration/registration.h, icp.h
//   pclhip::NormalEstimation, pclhip::VoxelGrid, pclhip::io::loadPCDFile / savePCDFile*
//
// With the stock device-backed parts plugged in (the default), align() runs the whole loop on the GPU
// (pclhip_icp_align); a foreign CorrespondenceEstimation or TransformationEstimation makes it run the generic
// loop of impl/icp.hpp:113-268 through the virtual calls.  With real PCL available the same C ABI sits inside
// subclasses of the real pcl:: bases: include/pclhip/pcl_plugin.hpp.
// Errors follow PCL's convention: no exceptions on the hot path, `false`/0 results + a message (getLastError()).
#pragma once

#include <cfloat>
#include <cmath>
#include <array>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <typeinfo>
#include <type_traits>
#include <vector>

#include "../pclhip.h"

namespace pclhip {

using index_t = std::int32_t;           // common/include/pcl/types.h:110-133
using Indices = std::vector<index_t>;
using IndicesPtr = std::shared_ptr<Indices>;
using IndicesConstPtr = std::shared_ptr<const Indices>;

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
// This header has exactly two point types, PointXYZ (16 bytes) and PointNormal (48 bytes: normal at +16, curvature at
// +32); their size tells them apart.  (The binding to real PCL classifies a type by its field list instead:
// pcl_plugin.hpp, record_layout.)
template <typename PointT> constexpr bool has_normal_fields() {
  static_assert(std::is_same<PointT, PointXYZ>::value || std::is_same<PointT, PointNormal>::value ||
                    std::is_same<PointT, Normal>::value,
                "pcl_compat.hpp: PointXYZ / PointNormal / Normal records only");
  return sizeof(PointT) >= 48;
}

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
  Correspondence() = default;
  Correspondence(index_t q, index_t m, float d) : index_query(q), index_match(m), distance(d) {}
};
using Correspondences = std::vector<Correspondence>;
using CorrespondencesPtr = std::shared_ptr<Correspondences>;

struct Matrix4f {  // row-major 4x4 with Eigen's coefficient access (row, col); PCL hands out Eigen::Matrix4f
  float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float& operator()(int r, int c) { return m[4 * r + c]; }
  float operator()(int r, int c) const { return m[4 * r + c]; }
  static Matrix4f Identity() { return Matrix4f(); }
  void setIdentity() { *this = Matrix4f(); }
  Matrix4f operator*(const Matrix4f& o) const {  // Eigen's coefficient order ((a0 b0 + a1 b1) + a2 b2) + a3 b3
    Matrix4f r;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        float s = m[4 * i] * o.m[j] + m[4 * i + 1] * o.m[4 + j];
        s = s + m[4 * i + 2] * o.m[8 + j];
        s = s + m[4 * i + 3] * o.m[12 + j];
        r.m[4 * i + j] = s;
      }
    return r;
  }
};

// One context per device, shared by the objects below (like PCL objects share nothing but the clouds).
class Context {
 public:
  using Ptr = std::shared_ptr<Context>;
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
  // the context objects use when none is given: PCL classes are default-constructible, so are these
  static Ptr defaultContext() {
    static Ptr c = std::make_shared<Context>(0);
    return c;
  }
 private:
  pclhip_ctx* ctx_ = nullptr;
  std::string error_;
};

// common/include/pcl/pcl_base.h:65-175
template <typename PointT>
class PCLBase {
 public:
  using PointCloudConstPtr = typename PointCloud<PointT>::ConstPtr;
  virtual ~PCLBase() = default;
  virtual void setInputCloud(const PointCloudConstPtr& cloud) { input_ = cloud; }
  PointCloudConstPtr getInputCloud() const { return input_; }
  virtual void setIndices(const IndicesPtr& indices) { indices_ = indices; fake_indices_ = false; }
  virtual void setIndices(const IndicesConstPtr& indices) { indices_ = std::make_shared<Indices>(*indices); fake_indices_ = false; }
  IndicesPtr getIndices() { return indices_; }
 protected:
  PointCloudConstPtr input_;
  IndicesPtr indices_;
  bool fake_indices_ = false;
  bool initCompute() {  // impl/pcl_base.hpp:138-174: no indices given -> all points
    if (!input_) return false;
    if (!indices_) { fake_indices_ = true; indices_ = std::make_shared<Indices>(); }
    if (fake_indices_ && indices_->size() != input_->size()) {
      indices_->resize(input_->size());
      for (std::size_t i = 0; i < indices_->size(); ++i) (*indices_)[i] = index_t(i);
    }
    return true;
  }
  bool usesAllPoints() const { return !indices_ || fake_indices_ || indices_->size() == input_->size(); }
};

// common/include/pcl/point_representation.h:59-190, 256-279, 546-579
template <typename PointT>
class PointRepresentation {
 public:
  using Ptr = std::shared_ptr<PointRepresentation<PointT>>;
  using ConstPtr = std::shared_ptr<const PointRepresentation<PointT>>;
  virtual ~PointRepresentation() = default;
  virtual void copyToFloatArray(const PointT& p, float* out) const = 0;
  template <typename OutputType> void vectorize(const PointT& p, OutputType& out) const {
    float t[16] = {0};
    copyToFloatArray(p, t);
    for (int i = 0; i < nr_dimensions_; ++i) out[i] = alpha_.empty() ? t[i] : t[i] * alpha_[std::size_t(i)];
  }
  void setRescaleValues(const float* rescale_array) { alpha_.assign(rescale_array, rescale_array + nr_dimensions_); }
  int getNumberOfDimensions() const { return nr_dimensions_; }
 protected:
  int nr_dimensions_ = 0;
  std::vector<float> alpha_;
};
template <typename PointT>
class DefaultPointRepresentation : public PointRepresentation<PointT> {
 public:
  DefaultPointRepresentation() { this->nr_dimensions_ = 3; }
  void copyToFloatArray(const PointT& p, float* out) const override { out[0] = p.x; out[1] = p.y; out[2] = p.z; }
};
template <typename PointT>
class CustomPointRepresentation : public PointRepresentation<PointT> {
 public:
  explicit CustomPointRepresentation(int max_dim = 3, int start_dim = 0) : start_dim_(start_dim) {
    this->nr_dimensions_ = max_dim < 16 ? max_dim : 16;
  }
  void copyToFloatArray(const PointT& p, float* out) const override {
    const float* f = reinterpret_cast<const float*>(&p) + start_dim_;
    for (int i = 0; i < this->nr_dimensions_; ++i) out[i] = f[i];
  }
 private:
  int start_dim_;
};

namespace search {

// pcl::search::Search<PointT>: what Feature::setSearchMethod and the registration classes talk to.  The batch
// overloads have the reference's default (a loop over the per-point virtual, search.hpp:113-136,164-190).
template <typename PointT>
class Search {
 public:
  using Ptr = std::shared_ptr<Search<PointT>>;
  using ConstPtr = std::shared_ptr<const Search<PointT>>;
  using PointCloudConstPtr = typename PointCloud<PointT>::ConstPtr;
  explicit Search(const std::string& name = "", bool sorted = false) : sorted_results_(sorted), name_(name) {}
  virtual ~Search() = default;
  virtual const std::string& getName() const { return name_; }
  virtual void setSortedResults(bool sorted) { sorted_results_ = sorted; }
  virtual bool getSortedResults() { return sorted_results_; }
  virtual bool setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) = 0;
  virtual PointCloudConstPtr getInputCloud() const { return input_; }
  virtual IndicesConstPtr getIndices() const { return indices_; }
  virtual int nearestKSearch(const PointT& point, int k, Indices& k_indices, std::vector<float>& k_sqr_distances) const = 0;
  virtual int nearestKSearch(const PointCloud<PointT>& cloud, index_t index, int k, Indices& k_indices,
                             std::vector<float>& k_sqr_distances) const {
    return nearestKSearch(cloud[std::size_t(index)], k, k_indices, k_sqr_distances);
  }
  virtual void nearestKSearch(const PointCloud<PointT>& cloud, const Indices& indices, int k, std::vector<Indices>& k_indices,
                              std::vector<std::vector<float>>& k_sqr_distances) const {
    const std::size_t n = indices.empty() ? cloud.size() : indices.size();
    k_indices.assign(n, Indices());
    k_sqr_distances.assign(n, std::vector<float>());
    for (std::size_t i = 0; i < n; ++i)
      nearestKSearch(cloud, indices.empty() ? index_t(i) : indices[i], k, k_indices[i], k_sqr_distances[i]);
  }
  virtual int radiusSearch(const PointT& point, double radius, Indices& k_indices, std::vector<float>& k_sqr_distances,
                           unsigned int max_nn = 0) const = 0;
  // queries of another point type (search.h:166-176,287-297): copied by x, y, z
  template <typename PointTDiff>
  int nearestKSearchT(const PointTDiff& point, int k, Indices& k_indices, std::vector<float>& k_sqr_distances) const {
    PointT p{};
    p.x = point.x; p.y = point.y; p.z = point.z;
    return nearestKSearch(p, k, k_indices, k_sqr_distances);
  }
  template <typename PointTDiff>
  int radiusSearchT(const PointTDiff& point, double radius, Indices& k_indices, std::vector<float>& k_sqr_distances,
                    unsigned int max_nn = 0) const {
    PointT p{};
    p.x = point.x; p.y = point.y; p.z = point.z;
    return radiusSearch(p, radius, k_indices, k_sqr_distances, max_nn);
  }
  // the device searches a whole batch per launch; host threads play no part (kept for source compatibility)
  virtual void setNumberOfThreads(unsigned int) {}
  virtual void radiusSearch(const PointCloud<PointT>& cloud, const Indices& indices, double radius,
                            std::vector<Indices>& k_indices, std::vector<std::vector<float>>& k_sqr_distances,
                            unsigned int max_nn = 0) const {
    const std::size_t n = indices.empty() ? cloud.size() : indices.size();
    k_indices.assign(n, Indices());
    k_sqr_distances.assign(n, std::vector<float>());
    for (std::size_t i = 0; i < n; ++i)
      radiusSearch(cloud[std::size_t(indices.empty() ? index_t(i) : indices[i])], radius, k_indices[i], k_sqr_distances[i], max_nn);
  }
 protected:
  PointCloudConstPtr input_;
  IndicesConstPtr indices_;
  bool sorted_results_;
  std::string name_;
};

// pcl::search::KdTree<PointT> on the device index: setInputCloud / nearestKSearch / radiusSearch (single +
// batch overloads; the batch ones are ONE launch).
template <typename PointT>
class KdTree : public Search<PointT> {
 public:
  using Ptr = std::shared_ptr<KdTree<PointT>>;
  using PointCloudConstPtr = typename PointCloud<PointT>::ConstPtr;
  using PointRepresentationConstPtr = typename PointRepresentation<PointT>::ConstPtr;
  using Search<PointT>::nearestKSearch;
  using Search<PointT>::radiusSearch;
  explicit KdTree(bool sorted = true) : KdTree(Context::defaultContext(), sorted) {}
  explicit KdTree(Context::Ptr ctx, bool sorted = true) : Search<PointT>("KdTree", sorted), ctx_(std::move(ctx)) {}
  ~KdTree() override { if (index_) pclhip_index_destroy(index_); }
  KdTree(const KdTree&) = delete;
  KdTree& operator=(const KdTree&) = delete;
  // search/kdtree.h:130-143, kdtree/kdtree.h:306-323: FLANN's eps allows a (1 + eps) approximate search; this index
  // always answers exactly, which satisfies every eps >= 0.  min_pts_ is stored for the getter (kdtree.h:325-338).
  void setEpsilon(float eps) { epsilon_ = eps; }
  float getEpsilon() const { return epsilon_; }
  void setMinPts(int min_pts) { min_pts_ = min_pts; }
  int getMinPts() const { return min_pts_; }

  // search/include/pcl/search/impl/kdtree.hpp:87-97 -> kdtree_flann.hpp:99-136: always (re)builds, like the
  // reference -- the cloud behind an unchanged pointer may have been modified in place.  Who knows it has not
  // keeps the tree and passes it with setSearchMethodTarget(tree, /*force_no_recompute=*/true).
  bool setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) override {
    if (index_) { pclhip_index_destroy(index_); index_ = nullptr; }
    this->input_ = cloud;
    this->indices_ = indices;
    if (!ctx_ || !ctx_->ok() || !cloud || unsupported_) return false;
    const bool sub = indices && !indices->empty();
    return pclhip_index_build_scaled(ctx_->get(), cloud->points.data(), sizeof(PointT), cloud->size(),
                                     sub ? indices->data() : nullptr, sub ? indices->size() : 0,
                                     scaled_ ? scale_ : nullptr, &index_) == PCLHIP_OK;
  }
  // kdtree.h:110: honoured for (x, y, z) prefixes with rescale values (the index is three-dimensional);
  // any other representation makes setInputCloud fail instead of searching something else
  void setPointRepresentation(const PointRepresentationConstPtr& rep) {
    rep_ = rep;
    scaled_ = unsupported_ = false;
    if (!rep) return;
    const int d = rep->getNumberOfDimensions();
    float probe[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    bool ok = d >= 1 && d <= 3;
    for (int a = 0; a < 3 && ok; ++a) {
      PointT p;
      p.x = a == 0 ? 1.0f : 0.0f; p.y = a == 1 ? 1.0f : 0.0f; p.z = a == 2 ? 1.0f : 0.0f;
      float out[16] = {0};
      rep->vectorize(p, out);
      for (int j = 0; j < d; ++j) probe[a][j] = out[j];
    }
    for (int a = 0; a < 3 && ok; ++a)
      for (int j = 0; j < 3; ++j)
        if (j != a && probe[a][j] != 0.0f) ok = false;
    if (!ok) { unsupported_ = true; return; }
    for (int a = 0; a < 3; ++a) scale_[a] = a < d ? probe[a][a] : 0.0f;
    scaled_ = !(scale_[0] == 1.0f && scale_[1] == 1.0f && scale_[2] == 1.0f);
    if (this->input_) setInputCloud(this->input_, this->indices_);
  }
  PointRepresentationConstPtr getPointRepresentation() const { return rep_; }
  bool hasDefaultRepresentation() const { return !scaled_ && !unsupported_; }

  // kdtree_flann.hpp:234-274: returns the number of neighbours found, resizes the outputs
  int nearestKSearch(const PointT& point, int k, Indices& k_indices, std::vector<float>& k_sqr_distances) const override {
    k_indices.clear();
    k_sqr_distances.clear();
    if (!index_ || k < 1) return 0;
    const std::uint64_t n = pclhip_index_size(index_);
    if (std::uint64_t(k) > n) k = int(n);  // :241-242
    if (k == 0) return 0;
    k_indices.resize(std::size_t(k));
    k_sqr_distances.resize(std::size_t(k));
    if (pclhip_knn(index_, &point, sizeof(PointT), 1, k, k_indices.data(), k_sqr_distances.data()) != PCLHIP_OK) return 0;
    int found = 0;
    while (found < k && k_indices[std::size_t(found)] >= 0) ++found;
    k_indices.resize(std::size_t(found));
    k_sqr_distances.resize(std::size_t(found));
    return found;
  }
  // batch overload, search/include/pcl/search/search.h:216-219 -- the efficient entry: one launch
  void nearestKSearch(const PointCloud<PointT>& cloud, const Indices& indices, int k, std::vector<Indices>& k_indices,
                      std::vector<std::vector<float>>& k_sqr_distances) const override {
    k_indices.clear();
    k_sqr_distances.clear();
    if (!index_ || k < 1) return;
    std::vector<PointT> q;
    const PointT* qp = cloud.points.data();
    std::size_t nq = cloud.size();
    if (!indices.empty()) {
      q.reserve(indices.size());
      for (index_t i : indices) q.push_back(cloud[std::size_t(i)]);
      qp = q.data();
      nq = q.size();
    }
    const std::uint64_t n = pclhip_index_size(index_);
    const int kk = std::uint64_t(k) > n ? int(n) : k;
    k_indices.assign(nq, Indices());
    k_sqr_distances.assign(nq, std::vector<float>());
    if (kk == 0 || nq == 0) return;
    Indices flat_i(nq * std::size_t(kk));
    std::vector<float> flat_d(nq * std::size_t(kk));
    if (pclhip_knn(index_, qp, sizeof(PointT), nq, kk, flat_i.data(), flat_d.data()) != PCLHIP_OK) return;
    for (std::size_t i = 0; i < nq; ++i) {
      int found = 0;
      while (found < kk && flat_i[i * std::size_t(kk) + std::size_t(found)] >= 0) ++found;
      k_indices[i].assign(flat_i.begin() + long(i * std::size_t(kk)), flat_i.begin() + long(i * std::size_t(kk)) + found);
      k_sqr_distances[i].assign(flat_d.begin() + long(i * std::size_t(kk)), flat_d.begin() + long(i * std::size_t(kk)) + found);
    }
  }
  // kdtree_flann.hpp:372-414: neighbours with squared distance < radius^2, ascending; max_nn = 0: all
  int radiusSearch(const PointT& point, double radius, Indices& k_indices, std::vector<float>& k_sqr_distances,
                   unsigned int max_nn = 0) const override {
    k_indices.clear();
    k_sqr_distances.clear();
    if (!index_) return 0;
    std::uint64_t off[2] = {0, 0}, total = 0;
    pclhip_status st = pclhip_radius_search(index_, &point, sizeof(PointT), 1, radius, max_nn, off, nullptr, nullptr, 0, &total);
    if ((st != PCLHIP_OK && st != PCLHIP_ERR_OVERFLOW) || total == 0) return 0;
    k_indices.resize(std::size_t(total));
    k_sqr_distances.resize(std::size_t(total));
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
                    unsigned int max_nn = 0) const override {
    k_indices.clear();
    k_sqr_distances.clear();
    if (!index_) return;
    std::vector<PointT> q;
    const PointT* qp = cloud.points.data();
    std::size_t nq = cloud.size();
    if (!indices.empty()) {
      for (index_t i : indices) q.push_back(cloud[std::size_t(i)]);
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
    Indices flat_i(static_cast<std::size_t>(total), 0);
    std::vector<float> flat_d(static_cast<std::size_t>(total), 0.0f);
    if (pclhip_radius_search(index_, qp, sizeof(PointT), nq, radius, max_nn, off.data(), flat_i.data(), flat_d.data(),
                             total, &total) != PCLHIP_OK) return;
    for (std::size_t i = 0; i < nq; ++i) {
      k_indices[i].assign(flat_i.begin() + long(off[i]), flat_i.begin() + long(off[i + 1]));
      k_sqr_distances[i].assign(flat_d.begin() + long(off[i]), flat_d.begin() + long(off[i + 1]));
    }
  }
  pclhip_index* handle() const { return index_; }
  Context::Ptr context() const { return ctx_; }

 private:
  Context::Ptr ctx_;
  pclhip_index* index_ = nullptr;
  PointRepresentationConstPtr rep_;
  float scale_[3] = {1, 1, 1};
  bool scaled_ = false, unsupported_ = false;
  float epsilon_ = 0.0f;
  int min_pts_ = 1;
};

}  // namespace search

// pcl::NormalEstimation<PointInT, pcl::Normal> (setKSearch or setRadiusSearch).
template <typename PointInT>
class NormalEstimation : public PCLBase<PointInT> {
 public:
  using SearchPtr = typename search::Search<PointInT>::Ptr;  // Feature::KdTreePtr is a search::Search (feature.h:119-120)
  NormalEstimation() : NormalEstimation(Context::defaultContext()) {}
  explicit NormalEstimation(Context::Ptr ctx) : ctx_(std::move(ctx)) {}
  void setSearchMethod(const SearchPtr& tree) { tree_ = tree; }
  SearchPtr getSearchMethod() const { return tree_; }
  void setKSearch(int k) { k_ = k; }
  int getKSearch() const { return k_; }
  void setRadiusSearch(double radius) { radius_ = radius; }
  double getRadiusSearch() const { return radius_; }
  double getSearchParameter() const { return k_ >= 1 ? double(k_) : radius_; }  // feature.h:190-197
  // Feature::setSearchSurface (features/include/pcl/features/feature.h:139-153): neighbours come from this cloud, one normal
  // per point of the input cloud (or per index, PCLBase::setIndices); unset = the input cloud itself
  void setSearchSurface(const typename PointCloud<PointInT>::ConstPtr& cloud) { surface_ = cloud; }
  typename PointCloud<PointInT>::ConstPtr getSearchSurface() const { return surface_; }
  void getViewPoint(float& x, float& y, float& z) const { x = vp_[0]; y = vp_[1]; z = vp_[2]; }
  // normal_3d.h:255-262 / :328-351: the cloud's sensor origin is the viewpoint until setViewPoint is called
  void setViewPoint(float x, float y, float z) { vp_[0] = x; vp_[1] = y; vp_[2] = z; use_sensor_origin_ = false; }
  void useSensorOriginAsViewPoint() { use_sensor_origin_ = true; }
  // Feature::compute (features/include/pcl/features/impl/feature.hpp:195-229); initCompute refuses
  // "both radius and K defined" and "neither defined" (:131-174).  The fused device kernel needs the device
  // search method (a foreign Search object has no neighbours-for-all-points entry this class could batch).
  void compute(PointCloud<Normal>& output) {
    output.points.clear();
    const auto& input = this->input_;
    if (!input || (k_ < 1) == !(radius_ > 0.0)) return;
    if (!tree_) tree_ = std::make_shared<search::KdTree<PointInT>>(ctx_);
    auto* dev = dynamic_cast<search::KdTree<PointInT>*>(tree_.get());
    if (dev == nullptr) return;
    const auto& surface = surface_ ? surface_ : input;  // feature.hpp:104-118
    if (dev->getInputCloud() != surface || dev->handle() == nullptr) {  // feature.hpp:125-130
      if (!dev->setInputCloud(surface)) return;
    }
    if (use_sensor_origin_) {
      vp_[0] = input->sensor_origin_[0]; vp_[1] = input->sensor_origin_[1]; vp_[2] = input->sensor_origin_[2];
    }
    const bool subset = this->indices_ && !this->fake_indices_;
    if (surface != input || subset) {  // every (selected) input point is a query against the surface
      const std::size_t m = subset ? this->indices_->size() : input->size();
      output.resize(m);
      std::uint64_t nan = 0;
      std::vector<float> tmp(m * 4);
      const pclhip_status st = pclhip_normals_at(dev->handle(), input->points.data(), sizeof(PointInT), input->size(),
                                                 subset ? this->indices_->data() : nullptr, subset ? m : 0, k_ >= 1 ? k_ : 0,
                                                 k_ >= 1 ? 0.0 : radius_, vp_, tmp.data(), 16, &nan);
      if (st != PCLHIP_OK) { output.points.clear(); return; }
      for (std::size_t i = 0; i < m; ++i) {
        output[i].normal_x = tmp[4 * i]; output[i].normal_y = tmp[4 * i + 1]; output[i].normal_z = tmp[4 * i + 2];
        output[i].curvature = tmp[4 * i + 3];
      }
      output.is_dense = (nan == 0);
      return;
    }
    output.resize(input->size());
    std::uint64_t nan = 0;
    static_assert(sizeof(Normal) == 32, "pcl::Normal: normal at +0, curvature at +16");
    // the output type of this class is pcl::Normal: whole records in one copy (pclhip_normals_records)
    if (pclhip_normals_records(dev->handle(), k_ >= 1 ? k_ : 0, k_ >= 1 ? 0.0 : radius_, vp_, output.points.data(), sizeof(Normal),
                               0, 16, &nan) != PCLHIP_OK) {
      output.points.clear();
      return;
    }
    output.is_dense = (nan == 0);  // normal_3d.hpp:56,63
  }
 private:
  Context::Ptr ctx_;
  SearchPtr tree_;
  typename PointCloud<PointInT>::ConstPtr surface_;
  int k_ = 0;
  double radius_ = 0.0;
  float vp_[3] = {0, 0, 0};
  bool use_sensor_origin_ = true;
};

// pcl::NormalEstimationOMP<PointInT, PointOutT> (features/include/pcl/features/normal_3d_omp.h:53-113): the same
// computation; the thread count of the host loop has no meaning for the one-launch device path and is only stored
template <typename PointInT>
class NormalEstimationOMP : public NormalEstimation<PointInT> {
 public:
  explicit NormalEstimationOMP(unsigned int nr_threads = 0) : NormalEstimation<PointInT>() { setNumberOfThreads(nr_threads); }
  explicit NormalEstimationOMP(Context::Ptr ctx, unsigned int nr_threads = 0) : NormalEstimation<PointInT>(std::move(ctx)) {
    setNumberOfThreads(nr_threads);
  }
  void setNumberOfThreads(unsigned int nr_threads = 0) { threads_ = nr_threads; }
  unsigned int getNumberOfThreads() const { return threads_; }
 private:
  unsigned int threads_ = 0;
};

namespace registration {

// pcl::registration::CorrespondenceRejector{Distance,MedianDistance,OneToOne,Trimmed}: parameter holders;
// the rejection itself runs on the device inside the ICP iteration (pclhip_icp_set_rejectors).
class CorrespondenceRejector {
 public:
  using Ptr = std::shared_ptr<CorrespondenceRejector>;
  pclhip_rejector desc{PCLHIP_REJ_DISTANCE, 0.0, 0, 0};
  virtual ~CorrespondenceRejector() = default;
  const std::string& getClassName() const { return rejection_name_; }
 protected:
  std::string rejection_name_;
};
struct CorrespondenceRejectorDistance : CorrespondenceRejector {
  CorrespondenceRejectorDistance() { desc.kind = PCLHIP_REJ_DISTANCE; rejection_name_ = "CorrespondenceRejectorDistance"; }
  void setMaximumDistance(float d) { desc.param = d; }  // correspondence_rejection_distance.h:93-97
  float getMaximumDistance() const { return float(desc.param); }
};
struct CorrespondenceRejectorMedianDistance : CorrespondenceRejector {
  CorrespondenceRejectorMedianDistance() {
    desc.kind = PCLHIP_REJ_MEDIAN_DISTANCE; desc.param = 1.0; rejection_name_ = "CorrespondenceRejectorMedianDistance";
  }
  void setMedianFactor(double f) { desc.param = f; }
  double getMedianFactor() const { return desc.param; }
};
struct CorrespondenceRejectorOneToOne : CorrespondenceRejector {
  CorrespondenceRejectorOneToOne() { desc.kind = PCLHIP_REJ_ONE_TO_ONE; rejection_name_ = "CorrespondenceRejectorOneToOne"; }
};
struct CorrespondenceRejectorTrimmed : CorrespondenceRejector {
  CorrespondenceRejectorTrimmed() { desc.kind = PCLHIP_REJ_TRIMMED; desc.param = 0.5; rejection_name_ = "CorrespondenceRejectorTrimmed"; }
  void setOverlapRatio(float r) { desc.param = r; }
  float getOverlapRatio() const { return float(desc.param); }
  void setMinCorrespondences(unsigned n) { desc.min_correspondences = n; }
  unsigned getMinCorrespondences() const { return desc.min_correspondences; }
};

// pcl::registration::TransformationEstimation (transformation_estimation.h:50-125): the four overloads
template <typename PointSource, typename PointTarget>
class TransformationEstimation {
 public:
  using Ptr = std::shared_ptr<TransformationEstimation<PointSource, PointTarget>>;
  using ConstPtr = std::shared_ptr<const TransformationEstimation<PointSource, PointTarget>>;
  using Matrix4 = Matrix4f;
  virtual ~TransformationEstimation() = default;
  virtual void estimateRigidTransformation(const PointCloud<PointSource>& cloud_src, const PointCloud<PointTarget>& cloud_tgt,
                                           Matrix4& transformation_matrix) const = 0;
  virtual void estimateRigidTransformation(const PointCloud<PointSource>& cloud_src, const Indices& indices_src,
                                           const PointCloud<PointTarget>& cloud_tgt, Matrix4& transformation_matrix) const = 0;
  virtual void estimateRigidTransformation(const PointCloud<PointSource>& cloud_src, const Indices& indices_src,
                                           const PointCloud<PointTarget>& cloud_tgt, const Indices& indices_tgt,
                                           Matrix4& transformation_matrix) const = 0;
  virtual void estimateRigidTransformation(const PointCloud<PointSource>& cloud_src, const PointCloud<PointTarget>& cloud_tgt,
                                           const Correspondences& correspondences, Matrix4& transformation_matrix) const = 0;
};

// The three estimators of the path on the device (pclhip_estimate_rigid_transformation): pair i = (src[i],
// tgt[i]) after the index lists / correspondences have been resolved (transformation_estimation_svd.hpp:49-125,
// ..._point_to_plane_lls.hpp:50-130).  Normals are read from the PointNormal layout (+16 bytes).  On failure
// (size mismatch, missing normals, no device) the matrix is left untouched, as the reference does.
template <typename PointSource, typename PointTarget, int MODE>
class DeviceTransformationEstimation : public TransformationEstimation<PointSource, PointTarget> {
 public:
  using Matrix4 = Matrix4f;
  DeviceTransformationEstimation() : ctx_(Context::defaultContext()) {}
  explicit DeviceTransformationEstimation(Context::Ptr ctx) : ctx_(std::move(ctx)) {}
  static constexpr int mode() { return MODE; }
  void setEnforceSameDirectionNormals(bool on) { enforce_ = on; }  // symmetric objective only
  bool getEnforceSameDirectionNormals() const { return enforce_; }
  void estimateRigidTransformation(const PointCloud<PointSource>& src, const PointCloud<PointTarget>& tgt,
                                   Matrix4& T) const override {
    if (src.size() != tgt.size()) return;  // "Number or points in source differs than target"
    run(src.points.data(), tgt.points.data(), src.size(), T);
  }
  void estimateRigidTransformation(const PointCloud<PointSource>& src, const Indices& indices_src,
                                   const PointCloud<PointTarget>& tgt, Matrix4& T) const override {
    if (indices_src.size() != tgt.size()) return;
    std::vector<PointSource> s;
    s.reserve(indices_src.size());
    for (index_t i : indices_src) s.push_back(src[std::size_t(i)]);
    run(s.data(), tgt.points.data(), s.size(), T);
  }
  void estimateRigidTransformation(const PointCloud<PointSource>& src, const Indices& indices_src,
                                   const PointCloud<PointTarget>& tgt, const Indices& indices_tgt, Matrix4& T) const override {
    if (indices_src.size() != indices_tgt.size()) return;
    std::vector<PointSource> s;
    std::vector<PointTarget> t;
    s.reserve(indices_src.size());
    t.reserve(indices_tgt.size());
    for (index_t i : indices_src) s.push_back(src[std::size_t(i)]);
    for (index_t i : indices_tgt) t.push_back(tgt[std::size_t(i)]);
    run(s.data(), t.data(), s.size(), T);
  }
  void estimateRigidTransformation(const PointCloud<PointSource>& src, const PointCloud<PointTarget>& tgt,
                                   const Correspondences& correspondences, Matrix4& T) const override {
    std::vector<PointSource> s;
    std::vector<PointTarget> t;
    s.reserve(correspondences.size());
    t.reserve(correspondences.size());
    for (const Correspondence& c : correspondences) {
      s.push_back(src[std::size_t(c.index_query)]);
      t.push_back(tgt[std::size_t(c.index_match)]);
    }
    run(s.data(), t.data(), s.size(), T);
  }
 private:
  void run(const PointSource* s, const PointTarget* t, std::size_t n, Matrix4& T) const {
    if (!ctx_ || !ctx_->ok()) return;
    const char* sb = reinterpret_cast<const char*>(s);
    const char* tb = reinterpret_cast<const char*>(t);
    const void* sn = (MODE == PCLHIP_ICP_SYMMETRIC && has_normal_fields<PointSource>()) ? sb + 16 : nullptr;
    const void* tn = (MODE != PCLHIP_ICP_POINT_TO_POINT && has_normal_fields<PointTarget>()) ? tb + 16 : nullptr;
    Matrix4 R;
    if (pclhip_estimate_rigid_transformation(ctx_->get(), MODE, sb, sizeof(PointSource), sn, sizeof(PointSource), tb,
                                             sizeof(PointTarget), tn, sizeof(PointTarget), n, enforce_ ? 1 : 0, R.m,
                                             nullptr) == PCLHIP_OK)
      T = R;
  }
  Context::Ptr ctx_;
  bool enforce_ = true;
};
template <typename S, typename T> using TransformationEstimationSVD = DeviceTransformationEstimation<S, T, PCLHIP_ICP_POINT_TO_POINT>;
template <typename S, typename T> using TransformationEstimationPointToPlaneLLS = DeviceTransformationEstimation<S, T, PCLHIP_ICP_POINT_TO_PLANE>;
template <typename S, typename T> using TransformationEstimationSymmetricPointToPlaneLLS = DeviceTransformationEstimation<S, T, PCLHIP_ICP_SYMMETRIC>;

// pcl::registration::CorrespondenceEstimationBase (correspondence_estimation.h:62-330)
template <typename PointSource, typename PointTarget>
class CorrespondenceEstimationBase : public PCLBase<PointSource> {
 public:
  using Ptr = std::shared_ptr<CorrespondenceEstimationBase<PointSource, PointTarget>>;
  using KdTree = search::KdTree<PointTarget>;
  using KdTreePtr = typename KdTree::Ptr;
  using PointCloudSourceConstPtr = typename PointCloud<PointSource>::ConstPtr;
  using PointCloudTargetConstPtr = typename PointCloud<PointTarget>::ConstPtr;
  void setInputSource(const PointCloudSourceConstPtr& cloud) { source_cloud_updated_ = true; PCLBase<PointSource>::setInputCloud(cloud); }
  PointCloudSourceConstPtr getInputSource() { return this->input_; }
  void setInputTarget(const PointCloudTargetConstPtr& cloud) {  // impl/correspondence_estimation.hpp:53-69
    if (!cloud || cloud->points.empty()) return;
    target_ = cloud;
    target_cloud_updated_ = true;
  }
  PointCloudTargetConstPtr getInputTarget() { return target_; }
  void setIndicesSource(const IndicesPtr& indices) { this->setIndices(indices); source_cloud_updated_ = true; }
  IndicesPtr getIndicesSource() { return this->indices_; }
  void setIndicesTarget(const IndicesPtr& indices) { target_cloud_updated_ = true; target_indices_ = indices; }
  IndicesPtr getIndicesTarget() { return target_indices_; }
  void setSearchMethodTarget(const KdTreePtr& tree, bool force_no_recompute = false) {
    tree_ = tree;
    force_no_recompute_ = force_no_recompute;
    target_cloud_updated_ = true;
  }
  KdTreePtr getSearchMethodTarget() const { return tree_; }
  // correspondence_estimation.h:230-262: the reciprocal search's tree over the source.  The device builds its own index of
  // the (moving) source every iteration, so a tree handed in here is kept for the getter only.
  using KdTreeReciprocal = search::KdTree<PointSource>;
  using KdTreeReciprocalPtr = typename KdTreeReciprocal::Ptr;
  void setSearchMethodSource(const KdTreeReciprocalPtr& tree, bool force_no_recompute = false) {
    tree_reciprocal_ = tree;
    force_no_recompute_reciprocal_ = force_no_recompute;
    source_cloud_updated_ = true;
  }
  KdTreeReciprocalPtr getSearchMethodSource() const { return tree_reciprocal_; }
  // :111-131: estimators that need normals say so; the nearest-neighbour estimator does not
  virtual bool requiresSourceNormals() const { return false; }
  virtual bool requiresTargetNormals() const { return false; }
  // :144-155 (OpenMP threads of the host loop): one device launch covers all points
  void setNumberOfThreads(unsigned int nr_threads) { num_threads_ = nr_threads; }
  virtual void determineCorrespondences(Correspondences& correspondences,
                                        double max_distance = std::numeric_limits<double>::max()) = 0;
  virtual void determineReciprocalCorrespondences(Correspondences& correspondences,
                                                  double max_distance = std::numeric_limits<double>::max()) = 0;
  virtual Ptr clone() const = 0;
 protected:
  KdTreePtr tree_;
  PointCloudTargetConstPtr target_;
  IndicesPtr target_indices_;
  KdTreeReciprocalPtr tree_reciprocal_;
  bool target_cloud_updated_ = true, source_cloud_updated_ = true, force_no_recompute_ = false;
  bool force_no_recompute_reciprocal_ = false;
  unsigned int num_threads_ = 1;
  bool initCompute() {  // impl/correspondence_estimation.hpp:71-97
    if (!target_ || !tree_) return false;
    if (target_cloud_updated_ && !force_no_recompute_) {
      if (!(target_indices_ ? tree_->setInputCloud(target_, target_indices_) : tree_->setInputCloud(target_))) return false;
      target_cloud_updated_ = false;
    }
    return PCLBase<PointSource>::initCompute();
  }
};

// pcl::registration::CorrespondenceEstimation: all source points in one launch
template <typename PointSource, typename PointTarget>
class CorrespondenceEstimation : public CorrespondenceEstimationBase<PointSource, PointTarget> {
  using Base = CorrespondenceEstimationBase<PointSource, PointTarget>;
 public:
  using Ptr = std::shared_ptr<CorrespondenceEstimation<PointSource, PointTarget>>;
  CorrespondenceEstimation() : CorrespondenceEstimation(Context::defaultContext()) {}
  explicit CorrespondenceEstimation(Context::Ptr ctx) : ctx_(std::move(ctx)) { this->tree_ = std::make_shared<search::KdTree<PointTarget>>(ctx_); }
  CorrespondenceEstimation(const CorrespondenceEstimation& o) : Base(o), ctx_(o.ctx_) {}  // the device handle is not shared
  ~CorrespondenceEstimation() override { if (icp_) pclhip_icp_destroy(icp_); }
  void determineCorrespondences(Correspondences& out, double max_distance = std::numeric_limits<double>::max()) override {
    run(out, max_distance, false);
  }
  void determineReciprocalCorrespondences(Correspondences& out, double max_distance = std::numeric_limits<double>::max()) override {
    run(out, max_distance, true);
  }
  typename Base::Ptr clone() const override { return std::make_shared<CorrespondenceEstimation<PointSource, PointTarget>>(*this); }
 private:
  void run(Correspondences& out, double max_distance, bool reciprocal) {
    out.clear();
    if (!this->initCompute() || this->tree_->handle() == nullptr) return;
    if (icp_ && icp_target_ != this->tree_->handle()) { pclhip_icp_destroy(icp_); icp_ = nullptr; }
    if (!icp_) {
      if (pclhip_icp_create(this->tree_->handle(), &icp_) != PCLHIP_OK) return;
      icp_target_ = this->tree_->handle();
    }
    const bool subset = !this->usesAllPoints();
    static const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double sums[PCLHIP_ICP_NSUMS];
    const double md = max_distance < 1e150 ? max_distance : 1e150;  // its square must stay finite
    if (pclhip_icp_set_source_indexed(icp_, this->input_->points.data(), sizeof(PointSource), this->input_->size(),
                                      subset ? this->indices_->data() : nullptr, subset ? this->indices_->size() : 0) == PCLHIP_OK &&
        pclhip_icp_set_reciprocal(icp_, reciprocal ? 1 : 0) == PCLHIP_OK &&
        pclhip_icp_iterate(icp_, I, md, PCLHIP_ICP_POINT_TO_POINT, sums) == PCLHIP_OK) {
      const std::size_t n = this->input_->size();
      std::uint64_t cnt = 0;
      if (!subset && sizeof(Correspondence) == 12) {  // the records compacted on the device, one copy
        out.resize(n);
        if (pclhip_icp_fetch_correspondence_records(icp_, out.data(), n, &cnt) == PCLHIP_OK) {
          out.resize(std::size_t(cnt));
          return;
        }
        out.clear();
      }
      Indices q(n), m(n);
      std::vector<float> d(n);
      if (pclhip_icp_fetch_correspondences(icp_, q.data(), m.data(), d.data(), &cnt) == PCLHIP_OK) {
        out.resize(std::size_t(cnt));
        for (std::uint64_t i = 0; i < cnt; ++i) out[std::size_t(i)] = Correspondence(q[std::size_t(i)], m[std::size_t(i)], d[std::size_t(i)]);
      }
    }
  }
  Context::Ptr ctx_;
  pclhip_icp* icp_ = nullptr;
  pclhip_index* icp_target_ = nullptr;
};

}  // namespace registration

// pcl::Registration<PointSource, PointTarget> (registration/include/pcl/registration/registration.h:56-700)
template <typename PointSource, typename PointTarget> class IterativeClosestPoint;

namespace registration {
// pcl::registration::DefaultConvergenceCriteria<Scalar> (default_convergence_criteria.h:61-286): the option holder a
// registration hands out through getConvergeCriteria(); hasConverged() itself runs on the device (closed_forms.hpp),
// its host twin is pclhip_convergence_has_converged.
class DefaultConvergenceCriteria {
 public:
  using Ptr = std::shared_ptr<DefaultConvergenceCriteria>;
  enum ConvergenceState {  // :73-81
    CONVERGENCE_CRITERIA_NOT_CONVERGED = 0,
    CONVERGENCE_CRITERIA_ITERATIONS = 1,
    CONVERGENCE_CRITERIA_TRANSFORM = 2,
    CONVERGENCE_CRITERIA_ABS_MSE = 3,
    CONVERGENCE_CRITERIA_REL_MSE = 4,
    CONVERGENCE_CRITERIA_NO_CORRESPONDENCES = 5,
    CONVERGENCE_CRITERIA_FAILURE_AFTER_MAX_ITERATIONS = 6
  };
  void setMaximumIterationsSimilarTransforms(int n) { max_iterations_similar_transforms_ = n; }
  int getMaximumIterationsSimilarTransforms() const { return max_iterations_similar_transforms_; }
  void setMaximumIterations(int n) { max_iterations_ = n; }
  int getMaximumIterations() const { return max_iterations_; }
  void setFailureAfterMaximumIterations(bool f) { failure_after_max_iter_ = f; }
  bool getFailureAfterMaximumIterations() const { return failure_after_max_iter_; }
  void setRotationThreshold(double t) { rotation_threshold_ = t; }
  double getRotationThreshold() const { return rotation_threshold_; }
  void setTranslationThreshold(double t) { translation_threshold_ = t; }
  double getTranslationThreshold() const { return translation_threshold_; }
  void setRelativeMSE(double m) { mse_threshold_relative_ = m; }
  double getRelativeMSE() const { return mse_threshold_relative_; }
  void setAbsoluteMSE(double m) { mse_threshold_absolute_ = m; }
  double getAbsoluteMSE() const { return mse_threshold_absolute_; }
  ConvergenceState getConvergenceState() const { return ConvergenceState(state_); }
  void setConvergenceState(ConvergenceState s) { state_ = int(s); }
 private:
  template <typename S, typename T> friend class ::pclhip::IterativeClosestPoint;
  int max_iterations_ = 100, max_iterations_similar_transforms_ = 0;  // :289-316
  bool failure_after_max_iter_ = false;
  double rotation_threshold_ = 0.99999, translation_threshold_ = 3e-4 * 3e-4;
  double mse_threshold_relative_ = 0.00001, mse_threshold_absolute_ = 1e-12;
  int state_ = 0;
};
}  // namespace registration

template <typename PointSource, typename PointTarget>
class Registration : public PCLBase<PointSource> {
 public:
  using Matrix4 = Matrix4f;
  using PointCloudSource = PointCloud<PointSource>;
  using PointCloudTarget = PointCloud<PointTarget>;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
  using KdTree = search::KdTree<PointTarget>;
  using KdTreePtr = typename KdTree::Ptr;
  using PointRepresentationConstPtr = typename PointRepresentation<PointTarget>::ConstPtr;
  using TransformationEstimation = registration::TransformationEstimation<PointSource, PointTarget>;
  using TransformationEstimationPtr = typename TransformationEstimation::Ptr;
  using CorrespondenceEstimation = registration::CorrespondenceEstimationBase<PointSource, PointTarget>;
  using CorrespondenceEstimationPtr = typename CorrespondenceEstimation::Ptr;
  using CorrespondenceRejectorPtr = registration::CorrespondenceRejector::Ptr;

  explicit Registration(Context::Ptr ctx)
      : ctx_(std::move(ctx)), tree_(std::make_shared<KdTree>(ctx_)), correspondences_(std::make_shared<Correspondences>()) {}
  void setTransformationEstimation(const TransformationEstimationPtr& te) { transformation_estimation_ = te; }  // :144-148
  void setCorrespondenceEstimation(const CorrespondenceEstimationPtr& ce) { correspondence_estimation_ = ce; }  // :173-177
  virtual void setInputSource(const PointCloudSourceConstPtr& cloud) {  // impl/registration.hpp:45-56: every call counts
    if (!cloud || cloud->points.empty()) return;
    source_cloud_updated_ = true;
    PCLBase<PointSource>::setInputCloud(cloud);
  }
  PointCloudSourceConstPtr getInputSource() { return this->input_; }
  virtual void setInputTarget(const PointCloudTargetConstPtr& cloud) {  // :58-69
    if (!cloud || cloud->points.empty()) return;
    target_ = cloud;
    target_cloud_updated_ = true;
  }
  PointCloudTargetConstPtr getInputTarget() { return target_; }
  void setIndices(const IndicesPtr& indices) override { PCLBase<PointSource>::setIndices(indices); source_cloud_updated_ = true; }
  void setIndices(const IndicesConstPtr& indices) override { PCLBase<PointSource>::setIndices(indices); source_cloud_updated_ = true; }
  void setSearchMethodTarget(const KdTreePtr& tree, bool force_no_recompute = false) {  // :214-221
    tree_ = tree;
    force_no_recompute_ = force_no_recompute;
    target_cloud_updated_ = true;
  }
  KdTreePtr getSearchMethodTarget() const { return tree_; }
  // registration.h:230-262 (the reciprocal search's source tree; see CorrespondenceEstimationBase::setSearchMethodSource)
  using KdTreeReciprocal = search::KdTree<PointSource>;
  using KdTreeReciprocalPtr = typename KdTreeReciprocal::Ptr;
  void setSearchMethodSource(const KdTreeReciprocalPtr& tree, bool force_no_recompute = false) {
    tree_reciprocal_ = tree;
    force_no_recompute_reciprocal_ = force_no_recompute;
    source_cloud_updated_ = true;
  }
  KdTreeReciprocalPtr getSearchMethodSource() const { return tree_reciprocal_; }
  // :300-322: RANSAC refinement parameters; IterativeClosestPoint does not use them (nor does the reference's)
  void setRANSACIterations(int n) { ransac_iterations_ = n; }
  double getRANSACIterations() const { return ransac_iterations_; }
  void setRANSACOutlierRejectionThreshold(double t) { inlier_threshold_ = t; }
  double getRANSACOutlierRejectionThreshold() const { return inlier_threshold_; }
  Matrix4 getFinalTransformation() { return final_transformation_; }
  Matrix4 getLastIncrementalTransformation() { return transformation_; }
  void setMaximumIterations(int nr_iterations) { max_iterations_ = nr_iterations; }
  int getMaximumIterations() const { return max_iterations_; }
  void setMaxCorrespondenceDistance(double d) { corr_dist_threshold_ = d; }
  double getMaxCorrespondenceDistance() const { return corr_dist_threshold_; }
  void setTransformationEpsilon(double e) { transformation_epsilon_ = e; }
  double getTransformationEpsilon() const { return transformation_epsilon_; }
  void setTransformationRotationEpsilon(double e) { transformation_rotation_epsilon_ = e; }
  double getTransformationRotationEpsilon() const { return transformation_rotation_epsilon_; }
  void setEuclideanFitnessEpsilon(double e) { euclidean_fitness_epsilon_ = e; }
  double getEuclideanFitnessEpsilon() const { return euclidean_fitness_epsilon_; }
  void setPointRepresentation(const PointRepresentationConstPtr& rep) { point_representation_ = rep; }  // :422
  void addCorrespondenceRejector(const CorrespondenceRejectorPtr& r) { correspondence_rejectors_.push_back(r); }  // :518-547
  std::vector<CorrespondenceRejectorPtr> getCorrespondenceRejectors() { return correspondence_rejectors_; }
  bool removeCorrespondenceRejector(unsigned int i) {
    if (i >= correspondence_rejectors_.size()) return false;
    correspondence_rejectors_.erase(correspondence_rejectors_.begin() + i);
    return true;
  }
  void clearCorrespondenceRejectors() { correspondence_rejectors_.clear(); }
  bool hasConverged() const { return converged_; }
  const std::string& getClassName() const { return reg_name_; }
  std::string getLastError() const { return ctx_->getLastError(); }

  void align(PointCloudSource& output) { align(output, Matrix4::Identity()); }
  void align(PointCloudSource& output, const Matrix4& guess) {  // impl/registration.hpp:178-221
    converged_ = false;
    if (!initCompute()) return;
    output.points.resize(this->indices_->size());
    for (std::size_t i = 0; i < this->indices_->size(); ++i) output[i] = (*this->input_)[std::size_t((*this->indices_)[i])];
    output.width = std::uint32_t(output.size());
    output.height = 1;
    output.is_dense = this->input_->is_dense;
    if (point_representation_ && !force_no_recompute_) tree_->setPointRepresentation(point_representation_);
    final_transformation_ = transformation_ = Matrix4::Identity();
    computeTransformation(output, guess);
  }
  virtual double getFitnessScore(double max_range = std::numeric_limits<double>::max()) = 0;

 protected:
  bool initCompute() {  // impl/registration.hpp:73-101
    if (!ctx_ || !ctx_->ok() || !this->input_ || !tree_) return false;
    if (target_cloud_updated_ && !(force_no_recompute_ && tree_->handle())) {
      if (!target_ || !tree_->setInputCloud(target_)) return false;
      target_built_ = true;
    }
    target_cloud_updated_ = false;
    if (!tree_->handle()) return false;
    if (correspondence_estimation_) correspondence_estimation_->setSearchMethodTarget(tree_, true);
    return PCLBase<PointSource>::initCompute();
  }
  virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) = 0;  // :678-679

  Context::Ptr ctx_;
  std::string reg_name_;
  KdTreePtr tree_;
  PointCloudTargetConstPtr target_;
  int nr_iterations_ = 0, max_iterations_ = 10;
  Matrix4 final_transformation_, transformation_;
  double transformation_epsilon_ = 0.0, transformation_rotation_epsilon_ = 0.0;
  double euclidean_fitness_epsilon_ = -std::numeric_limits<double>::max();
  double corr_dist_threshold_ = std::sqrt(std::numeric_limits<double>::max());
  bool converged_ = false;
  int min_number_correspondences_ = 3;
  CorrespondencesPtr correspondences_;
  TransformationEstimationPtr transformation_estimation_;
  CorrespondenceEstimationPtr correspondence_estimation_;
  std::vector<CorrespondenceRejectorPtr> correspondence_rejectors_;
  PointRepresentationConstPtr point_representation_;
  bool target_cloud_updated_ = true, source_cloud_updated_ = true, force_no_recompute_ = false;
  bool target_built_ = false;  // the tree was (re)built by this object since the device handle was last refreshed
  KdTreeReciprocalPtr tree_reciprocal_;
  bool force_no_recompute_reciprocal_ = false;
  int ransac_iterations_ = 0;       // registration.h:570
  double inlier_threshold_ = 0.05;  // :600
};

// pcl::IterativeClosestPoint<PointSource, PointTarget> (icp.h:98-347)
template <typename PointSource, typename PointTarget>
class IterativeClosestPoint : public Registration<PointSource, PointTarget> {
  using Base = Registration<PointSource, PointTarget>;
 public:
  using PointCloudSource = typename Base::PointCloudSource;
  using Matrix4 = typename Base::Matrix4;
  using Ptr = std::shared_ptr<IterativeClosestPoint<PointSource, PointTarget>>;
  IterativeClosestPoint() : IterativeClosestPoint(Context::defaultContext()) {}
  explicit IterativeClosestPoint(Context::Ptr ctx) : Base(std::move(ctx)) {  // icp.h:136-151
    this->reg_name_ = "IterativeClosestPoint";
    this->transformation_estimation_ = std::make_shared<registration::TransformationEstimationSVD<PointSource, PointTarget>>(this->ctx_);
    this->correspondence_estimation_ = std::make_shared<registration::CorrespondenceEstimation<PointSource, PointTarget>>(this->ctx_);
    convergence_criteria_ = std::make_shared<registration::DefaultConvergenceCriteria>();  // icp.h:144-146
    pclhip_convergence_init(&criteria_);
  }
  ~IterativeClosestPoint() override { if (icp_) pclhip_icp_destroy(icp_); }
  IterativeClosestPoint(const IterativeClosestPoint&) = delete;             // icp.h:168-173
  IterativeClosestPoint& operator=(const IterativeClosestPoint&) = delete;

  void setUseReciprocalCorrespondences(bool on) { use_reciprocal_correspondence_ = on; }  // icp.h:251-256
  bool getUseReciprocalCorrespondences() const { return use_reciprocal_correspondence_; }
  // icp.h:180-184: the criteria object; its similar-transforms count, failure-after-max-iterations flag and absolute MSE
  // threshold are the options a caller sets there (the other thresholds are overwritten from the registration's own
  // setters at every align(), impl/icp.hpp:157-161)
  registration::DefaultConvergenceCriteria::Ptr getConvergeCriteria() { return convergence_criteria_; }
  // shortcuts for the same three options
  void setMaximumIterationsSimilarTransforms(int n) { convergence_criteria_->setMaximumIterationsSimilarTransforms(n); }
  void setFailureAfterMaximumIterations(bool f) { convergence_criteria_->setFailureAfterMaximumIterations(f); }
  void setAbsoluteMSE(double mse) { convergence_criteria_->setAbsoluteMSE(mse); }
  int getNumberOfIterations() const { return this->nr_iterations_; }
  int getConvergenceState() const { return criteria_.convergence_state; }
  double getLastMSE() const { return last_mse_; }
  // true when the last align() ran the fused device loop, false when it went through the virtual calls
  bool ranOnDeviceLoop() const { return device_loop_; }

  // Registration::getFitnessScore (registration/include/pcl/registration/impl/registration.hpp:132-168)
  double getFitnessScore(double max_range = std::numeric_limits<double>::max()) override {
    if (!this->initCompute() || !ensureHandle(modeOfEstimator() >= 0 ? modeOfEstimator() : PCLHIP_ICP_POINT_TO_POINT))
      return std::numeric_limits<double>::max();
    double score = std::numeric_limits<double>::max();
    pclhip_icp_fitness_score(icp_, this->final_transformation_.m, max_range, &score, nullptr);
    return score;
  }

 protected:
  virtual bool enforceSameDirectionNormals() const { return true; }
  // which stock estimator is plugged in (its device mode), or -1 for a foreign one
  int modeOfEstimator() const {
    using namespace registration;
    const auto* te = this->transformation_estimation_.get();
    if (te == nullptr) return -1;
    // exactly the stock classes: a subclass that overrides them is a foreign estimator like any other
    if (typeid(*te) == typeid(TransformationEstimationSymmetricPointToPlaneLLS<PointSource, PointTarget>)) return PCLHIP_ICP_SYMMETRIC;
    if (typeid(*te) == typeid(TransformationEstimationPointToPlaneLLS<PointSource, PointTarget>)) return PCLHIP_ICP_POINT_TO_PLANE;
    if (typeid(*te) == typeid(TransformationEstimationSVD<PointSource, PointTarget>)) return PCLHIP_ICP_POINT_TO_POINT;
    return -1;
  }
  void fillParams(pclhip_icp_params& p, int mode) const {
    pclhip_icp_params_default(&p);
    p.mode = mode;
    p.max_iterations = this->max_iterations_;
    p.max_correspondence_distance = this->corr_dist_threshold_;
    p.transformation_epsilon = this->transformation_epsilon_;
    p.transformation_rotation_epsilon = this->transformation_rotation_epsilon_;
    p.euclidean_fitness_epsilon = this->euclidean_fitness_epsilon_;
    p.min_number_correspondences = this->min_number_correspondences_;
    auto& cc = *convergence_criteria_;
    p.failure_after_max_iterations = cc.getFailureAfterMaximumIterations() ? 1 : 0;
    p.max_iterations_similar_transforms = cc.getMaximumIterationsSimilarTransforms();
    p.mse_threshold_absolute = cc.getAbsoluteMSE();
    // impl/icp.hpp:157-161: what the loop pushes into the criteria before it starts
    cc.setMaximumIterations(this->max_iterations_);
    cc.setRelativeMSE(this->euclidean_fitness_epsilon_);
    cc.setTranslationThreshold(this->transformation_epsilon_);
    if (this->transformation_rotation_epsilon_ > 0) cc.setRotationThreshold(this->transformation_rotation_epsilon_);
    else cc.setRotationThreshold(0.99999);
  }
  // device-side registration object bound to the tree; source (and normals) uploaded when they changed
  bool ensureHandle(int mode) {
    pclhip_index* ix = this->tree_->handle();
    if (icp_ && (icp_target_ != ix || this->target_built_)) { pclhip_icp_destroy(icp_); icp_ = nullptr; }
    if (mode != PCLHIP_ICP_POINT_TO_POINT && has_normal_fields<PointTarget>() && this->target_ &&
        (this->target_built_ || normals_of_ != this->target_.get())) {  // pcl::PointNormal target: normals at +16
      const char* base = reinterpret_cast<const char*>(this->target_->points.data());
      if (pclhip_index_set_normals(ix, base + 16, sizeof(PointTarget)) != PCLHIP_OK) return false;
      normals_of_ = this->target_.get();
    }
    this->target_built_ = false;
    if (!icp_) {
      if (pclhip_icp_create(ix, &icp_) != PCLHIP_OK) return false;
      icp_target_ = ix;
      this->source_cloud_updated_ = true;
    }
    if (this->source_cloud_updated_) {
      const bool subset = !this->usesAllPoints();
      if (pclhip_icp_set_source_indexed(icp_, this->input_->points.data(), sizeof(PointSource), this->input_->size(),
                                        subset ? this->indices_->data() : nullptr, subset ? this->indices_->size() : 0) != PCLHIP_OK)
        return false;
      if (has_normal_fields<PointSource>()) {  // pcl::PointNormal source: its normals feed the symmetric objective
        const char* base = reinterpret_cast<const char*>(this->input_->points.data());
        if (pclhip_icp_set_source_normals(icp_, base + 16, sizeof(PointSource)) != PCLHIP_OK) return false;
      }
      this->source_cloud_updated_ = false;
    }
    std::vector<pclhip_rejector> list;
    for (const auto& r : this->correspondence_rejectors_) list.push_back(r->desc);
    return pclhip_icp_set_rejectors(icp_, list.data(), int(list.size())) == PCLHIP_OK &&
           pclhip_icp_set_reciprocal(icp_, use_reciprocal_correspondence_ ? 1 : 0) == PCLHIP_OK &&
           pclhip_icp_set_enforce_same_direction_normals(icp_, enforceSameDirectionNormals() ? 1 : 0) == PCLHIP_OK;
  }
  // IterativeClosestPoint::transformCloud (impl/icp.hpp:49-111), in place on a host cloud
  virtual void transformCloud(PointCloudSource& cloud, const Matrix4& T, int mode) {
    pclhip_transform_cloud(this->ctx_->get(), T.m, mode == PCLHIP_ICP_POINT_TO_POINT ? 0 : 1, cloud.points.data(),
                           cloud.points.data(), sizeof(PointSource), cloud.size(),
                           (mode != PCLHIP_ICP_POINT_TO_POINT && has_normal_fields<PointSource>()) ? 16 : 0);
  }

  void computeTransformation(PointCloudSource& output, const Matrix4& guess) override {  // impl/icp.hpp:113-268
    using namespace registration;
    const int mode = modeOfEstimator();
    const auto* cep = this->correspondence_estimation_.get();
    const bool own_ce = cep != nullptr && typeid(*cep) == typeid(registration::CorrespondenceEstimation<PointSource, PointTarget>);
    device_loop_ = mode >= 0 && own_ce && this->tree_->hasDefaultRepresentation();
    if (device_loop_) {
      if (!ensureHandle(mode)) return;
      pclhip_icp_params p;
      fillParams(p, mode);
      pclhip_icp_result r;
      if (pclhip_icp_align(icp_, &p, guess.m, &r) != PCLHIP_OK) return;
      std::memcpy(this->final_transformation_.m, r.final_transformation, sizeof r.final_transformation);
      std::memcpy(this->transformation_.m, r.last_transformation, sizeof r.last_transformation);
      this->converged_ = r.converged != 0;
      this->nr_iterations_ = r.nr_iterations;
      criteria_.convergence_state = r.convergence_state;
      convergence_criteria_->setConvergenceState(registration::DefaultConvergenceCriteria::ConvergenceState(r.convergence_state));
      last_mse_ = r.mse;
      output = *this->input_;  // icp.hpp:264-267: all fields of the WHOLE input cloud, then xyz (+ normals) moved
      transformCloud(output, this->final_transformation_, mode);
      return;
    }
    // ---- a foreign estimator is plugged in: PCL's loop, through the virtual interfaces -------------------
    if (!this->target_ || !this->transformation_estimation_ || !this->correspondence_estimation_) return;
    const int tmode = mode >= 0 ? mode : (has_normal_fields<PointSource>() ? PCLHIP_ICP_POINT_TO_PLANE : PCLHIP_ICP_POINT_TO_POINT);
    auto moved = std::make_shared<PointCloudSource>(output);  // input_transformed (:120-131), the indexed subset
    this->nr_iterations_ = 0;
    this->converged_ = false;
    this->final_transformation_ = guess;
    bool identity = true;
    for (int i = 0; i < 16; ++i) identity = identity && guess.m[i] == Matrix4().m[i];
    if (!identity) transformCloud(*moved, guess, tmode);
    this->transformation_ = Matrix4::Identity();
    auto& ce = *this->correspondence_estimation_;
    ce.setInputTarget(this->target_);  // :145 (the tree itself was handed over in initCompute)
    ce.setSearchMethodTarget(this->tree_, true);
    pclhip_icp_params p;
    fillParams(p, tmode);
    if (!this->correspondence_rejectors_.empty()) return;  // the rejectors here are device-side parameter holders
    do {
      ce.setInputSource(moved);  // :178
      if (use_reciprocal_correspondence_) ce.determineReciprocalCorrespondences(*this->correspondences_, this->corr_dist_threshold_);
      else ce.determineCorrespondences(*this->correspondences_, this->corr_dist_threshold_);
      const std::size_t cnt = this->correspondences_->size();
      if (int(cnt) < this->min_number_correspondences_) {  // :204-213
        criteria_.convergence_state = 5;  // CONVERGENCE_CRITERIA_NO_CORRESPONDENCES
        this->converged_ = false;
        break;
      }
      this->transformation_estimation_->estimateRigidTransformation(*moved, *this->target_, *this->correspondences_,
                                                                    this->transformation_);  // :216-217
      transformCloud(*moved, this->transformation_, tmode);                                     // :220
      this->final_transformation_ = this->transformation_ * this->final_transformation_;        // :223
      ++this->nr_iterations_;
      double mse = 0.0;  // calculateMSE, default_convergence_criteria.h:262-270
      for (const Correspondence& c : *this->correspondences_) mse += double(c.distance);
      mse /= double(cnt);
      last_mse_ = mse;
      this->converged_ = pclhip_convergence_has_converged(&p, &criteria_, this->nr_iterations_, this->transformation_.m, mse) != 0;
    } while (criteria_.convergence_state == 0);
    convergence_criteria_->setConvergenceState(registration::DefaultConvergenceCriteria::ConvergenceState(criteria_.convergence_state));
    output = *this->input_;
    transformCloud(output, this->final_transformation_, tmode);
  }

  bool use_reciprocal_correspondence_ = false;
  registration::DefaultConvergenceCriteria::Ptr convergence_criteria_;
  pclhip_convergence_state criteria_;
  double last_mse_ = 0;
  bool device_loop_ = false;
  pclhip_icp* icp_ = nullptr;
  pclhip_index* icp_target_ = nullptr;
  const void* normals_of_ = nullptr;
};

// pcl::IterativeClosestPointWithNormals (icp.h:360-440)
template <typename PointSource, typename PointTarget>
class IterativeClosestPointWithNormals : public IterativeClosestPoint<PointSource, PointTarget> {
 public:
  using Ptr = std::shared_ptr<IterativeClosestPointWithNormals<PointSource, PointTarget>>;
  IterativeClosestPointWithNormals() : IterativeClosestPointWithNormals(Context::defaultContext()) {}
  explicit IterativeClosestPointWithNormals(Context::Ptr ctx) : IterativeClosestPoint<PointSource, PointTarget>(std::move(ctx)) {
    this->reg_name_ = "IterativeClosestPointWithNormals";
    setUseSymmetricObjective(false);
  }
  void setUseSymmetricObjective(bool on) {  // icp.h:380-400
    use_symmetric_objective_ = on;
    if (on) {
      auto te = std::make_shared<registration::TransformationEstimationSymmetricPointToPlaneLLS<PointSource, PointTarget>>(this->ctx_);
      te->setEnforceSameDirectionNormals(enforce_same_direction_normals_);
      this->transformation_estimation_ = te;
    } else {
      this->transformation_estimation_ =
          std::make_shared<registration::TransformationEstimationPointToPlaneLLS<PointSource, PointTarget>>(this->ctx_);
    }
  }
  bool getUseSymmetricObjective() const { return use_symmetric_objective_; }
  void setEnforceSameDirectionNormals(bool on) {  // icp.h:416-428
    enforce_same_direction_normals_ = on;
    if (use_symmetric_objective_) setUseSymmetricObjective(true);
  }
  bool getEnforceSameDirectionNormals() const { return enforce_same_direction_normals_; }
 protected:
  bool enforceSameDirectionNormals() const override { return enforce_same_direction_normals_; }
  bool use_symmetric_objective_ = false, enforce_same_direction_normals_ = true;
};

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

// pcl::VoxelGrid<PointT> (filters/include/pcl/filters/voxel_grid.h:210-533) for pcl::PointXYZ and pcl::PointNormal
template <typename PointT = PointXYZ>
class VoxelGrid {
 public:
  VoxelGrid() : VoxelGrid(Context::defaultContext()) {}
  explicit VoxelGrid(Context::Ptr ctx) : ctx_(std::move(ctx)) {}
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { input_ = c; }
  typename PointCloud<PointT>::ConstPtr getInputCloud() const { return input_; }
  void setLeafSize(float lx, float ly, float lz) { leaf_[0] = lx; leaf_[1] = ly; leaf_[2] = lz; }
  std::array<float, 3> getLeafSize() const { return {leaf_[0], leaf_[1], leaf_[2]}; }
  // :293-301: false -> only x, y, z are averaged, every other field of the output keeps its default
  void setDownsampleAllData(bool downsample) { downsample_all_data_ = downsample; }
  bool getDownsampleAllData() const { return downsample_all_data_; }
  void setMinimumPointsNumberPerVoxel(unsigned n) { min_pts_ = n; }
  unsigned getMinimumPointsNumberPerVoxel() const { return min_pts_; }
  // :440-476: pass-through filter on one field of the point type before the grid is laid out
  void setFilterFieldName(const std::string& f) { field_ = f; }
  const std::string& getFilterFieldName() const { return field_; }
  void setFilterLimits(double lo, double hi) { lo_ = lo; hi_ = hi; }
  void getFilterLimits(double& lo, double& hi) const { lo = lo_; hi = hi_; }
  void setFilterLimitsNegative(bool negative) { negative_ = negative; }  // true: keep what lies OUTSIDE the interval
  bool getFilterLimitsNegative() const { return negative_; }
  // Filter::filter -> applyFilter (filters/include/pcl/filters/impl/voxel_grid.hpp:597-814); when the
  // voxel index would overflow the reference warns and returns the input unchanged (:620-629)
  void filter(PointCloud<PointT>& output) {
    output.points.clear();
    output.height = 1;
    output.is_dense = true;
    if (!input_) { output.width = 0; return; }
    const int limits = limitFlags();
    if (limits < 0) { output.width = 0; return; }  // "could not find field": refused, not silently ignored
    constexpr std::size_t noff = has_normal_fields<PointT>() ? 16 : 0;
    std::vector<PointT> out(input_->size());
    std::uint64_t n = 0;
    leaf_layout_.clear();
    dims_ = pclhip_voxelgrid_dims{};
    if (save_leaf_layout_ && !input_->empty()) {  // one int per grid cell: learn the grid first (bounding-box pass)
      if (pclhip_voxelgrid_grid(ctx_->get(), input_->points.data(), sizeof(PointT), input_->size(), leaf_, limits,
                                lo_, hi_, &dims_) == PCLHIP_OK)
        leaf_layout_.assign(std::size_t(dims_.div_b[0]) * std::size_t(dims_.div_b[1]) * std::size_t(dims_.div_b[2]), -1);
    }
    const pclhip_status st = pclhip_voxelgrid_ex2(ctx_->get(), input_->points.data(), sizeof(PointT), input_->size(), leaf_,
                                                  min_pts_, limits, lo_, hi_, downsample_all_data_ ? 1 : 0, noff,
                                                  out.data(), sizeof(PointT), &n,
                                                  leaf_layout_.empty() ? nullptr : leaf_layout_.data(), leaf_layout_.size(),
                                                  &dims_);
    if (st == PCLHIP_ERR_OVERFLOW) { output = *input_; return; }
    if (st != PCLHIP_OK) { output.width = 0; return; }
    out.resize(std::size_t(n));
    output.points.swap(out);
    output.width = std::uint32_t(n);
  }
  // the grid of the last filter() and the leaf layout (filters/include/pcl/filters/voxel_grid.h:316-421)
  void setSaveLeafLayout(bool save) { save_leaf_layout_ = save; }
  bool getSaveLeafLayout() const { return save_leaf_layout_; }
  std::array<int, 3> getMinBoxCoordinates() const { return {dims_.min_b[0], dims_.min_b[1], dims_.min_b[2]}; }
  std::array<int, 3> getMaxBoxCoordinates() const { return {dims_.max_b[0], dims_.max_b[1], dims_.max_b[2]}; }
  std::array<int, 3> getNrDivisions() const { return {dims_.div_b[0], dims_.div_b[1], dims_.div_b[2]}; }
  std::array<int, 3> getDivisionMultiplier() const { return {dims_.divb_mul[0], dims_.divb_mul[1], dims_.divb_mul[2]}; }
  std::vector<int> getLeafLayout() const { return std::vector<int>(leaf_layout_.begin(), leaf_layout_.end()); }
  std::array<int, 3> getGridCoordinates(float x, float y, float z) const {
    return {int(std::floor(x * (1.0f / leaf_[0]))), int(std::floor(y * (1.0f / leaf_[1]))), int(std::floor(z * (1.0f / leaf_[2])))};
  }
  int getCentroidIndexAt(const std::array<int, 3>& ijk) const {
    long long idx = 0;
    for (int d = 0; d < 3; ++d) idx += (long long)(ijk[d] - dims_.min_b[d]) * dims_.divb_mul[d];
    if (idx < 0 || idx >= (long long)leaf_layout_.size()) return -1;
    return leaf_layout_[std::size_t(idx)];
  }
  int getCentroidIndex(const PointT& p) const { return getCentroidIndexAt(getGridCoordinates(p.x, p.y, p.z)); }
  // :353-376: centroid indices of the cells at the given offsets around the reference point's cell
  std::vector<int> getNeighborCentroidIndices(const PointT& reference_point, const std::vector<std::array<int, 3>>& relative_coordinates) const {
    const std::array<int, 3> c = getGridCoordinates(reference_point.x, reference_point.y, reference_point.z);
    std::vector<int> neighbors;
    for (const auto& r : relative_coordinates) {
      const std::array<int, 3> cell = {c[0] + r[0], c[1] + r[1], c[2] + r[2]};
      bool inside = true;  // the reference clamps by min_b_/max_b_
      for (int d = 0; d < 3; ++d) inside = inside && cell[d] >= dims_.min_b[d] && cell[d] <= dims_.max_b[d];
      neighbors.push_back(inside ? getCentroidIndexAt(cell) : -1);
    }
    return neighbors;
  }
 private:
  // the `has_z_limits` word of the C ABI: where the named field sits in PointT (point_types.hpp:315-321, 843-853)
  int limitFlags() const {
    if (field_.empty()) return 0;
    int idx = -1;
    if (field_ == "x") idx = 0;
    else if (field_ == "y") idx = 1;
    else if (field_ == "z") idx = 2;
    else if (has_normal_fields<PointT>()) {
      if (field_ == "normal_x") idx = 4;
      else if (field_ == "normal_y") idx = 5;
      else if (field_ == "normal_z") idx = 6;
      else if (field_ == "curvature") idx = 8;
    }
    return idx < 0 ? -1 : PCLHIP_VOXELGRID_LIMITS(idx, negative_);
  }
  Context::Ptr ctx_;
  typename PointCloud<PointT>::ConstPtr input_;
  bool save_leaf_layout_ = false, downsample_all_data_ = true;  // voxel_grid.h:501
  bool negative_ = false;
  std::vector<std::int32_t> leaf_layout_;
  pclhip_voxelgrid_dims dims_{};
  float leaf_[3] = {0, 0, 0};
  unsigned min_pts_ = 0;
  std::string field_;
  double lo_ = -FLT_MAX, hi_ = FLT_MAX;
};

}  // namespace pclhip
