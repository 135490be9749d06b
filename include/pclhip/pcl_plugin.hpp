// pcl_plugin.hpp -- the binding that plugs libpclhip.so INTO real PCL objects: subclasses of PCL's own virtual
// plugin points, written against PCL's public headers only.  A PCL application switches by constructing these
// instead of the stock classes; everything else (setters, getters, align(), the containers) is PCL's own code.
//
//   pclhip::plugin::KdTreeHIP<PointT>                    : pcl::search::KdTree<PointT>
//        search/include/pcl/search/kdtree.h:61-168 -- what Registration::setSearchMethodTarget,
//        CorrespondenceEstimationBase::setSearchMethodTarget and Feature::setSearchMethod accept
//   pclhip::plugin::CorrespondenceEstimationHIP<S,T>     : pcl::registration::CorrespondenceEstimationBase<S,T,float>
//        registration/include/pcl/registration/correspondence_estimation.h:62-330 -- the BATCH entry
//        (Registration::setCorrespondenceEstimation, registration.h:173-177)
//   pclhip::plugin::IterativeClosestPointHIP<S,T>        : pcl::IterativeClosestPoint<S,T,float>
//   pclhip::plugin::IterativeClosestPointWithNormalsHIP  : pcl::IterativeClosestPointWithNormals<S,T,float>
//        override the protected computeTransformation (registration.h:678-679, icp.h:292-293): the whole loop
//        of impl/icp.hpp:113-268 runs on the device
//
//   pclhip::plugin::NormalEstimationHIP<In,Out>          : pcl::NormalEstimation<In,Out>
//        overrides computeFeature (features/include/pcl/features/normal_3d.h:398-399), like NormalEstimationOMP
//        (normal_3d_omp.h:53): k / radius, search surface, indices and view point stay PCL's own members
//   pclhip::plugin::VoxelGridHIP<PointT>                 : pcl::VoxelGrid<PointT>
//        overrides applyFilter (filters/include/pcl/filters/voxel_grid.h:529-530): leaf size, field filter,
//        downsample_all_data, minimum points and the leaf layout stay PCL's own members
//   pclhip::plugin::TransformationEstimationHIP<S,T,MODE> : pcl::registration::TransformationEstimation<S,T,float>
//        the four overloads of transformation_estimation.h:74-116 on explicit pairs
//
// Compiled and run here against tests/cpp/pcl_mock (a stand-in for the PCL base classes with the same
// signatures; PCL itself needs Eigen/Boost/FLANN, which this image lacks): tests/cpp/test_pcl_plugin.cpp.
// Matrices cross the boundary through operator()(row, col) only, so Eigen's storage order does not matter.
#pragma once

#include <pcl/common/io.h>
#include <pcl/features/normal_3d.h>
#include <pcl/filters/voxel_grid.h>
#include <pcl/registration/correspondence_estimation.h>
#include <pcl/registration/icp.h>
#include <pcl/registration/transformation_estimation.h>
#include <pcl/search/kdtree.h>

#include <chrono>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "../pclhip.h"

namespace pclhip {
namespace plugin {

// One device context shared by the objects of an application (pclhip_ctx).
class Device {
 public:
  using Ptr = std::shared_ptr<Device>;
  explicit Device(int device = 0, void* hip_stream = nullptr) {
    if (pclhip_ctx_create(device, hip_stream, &ctx_) != PCLHIP_OK) ctx_ = nullptr;
  }
  ~Device() { if (ctx_) pclhip_ctx_destroy(ctx_); }
  Device(const Device&) = delete;
  Device& operator=(const Device&) = delete;
  pclhip_ctx* get() const { return ctx_; }
  bool ok() const { return ctx_ != nullptr; }
  static Ptr instance() {  // the default device of the process
    static Ptr d = std::make_shared<Device>(0);
    return d;
  }
 private:
  pclhip_ctx* ctx_ = nullptr;
};

// What the device kernels know of a record: (x, y, z) floats at +0, and of PointNormal's extras the normal at +16 and
// the curvature at +32.  A point type is classified by its field list (common/include/pcl/common/io.h:78-96), not by
// its size: PointXYZI / PointXYZRGB carry fields the device path would drop, and the 48-byte PointXYZRGBNormal /
// PointXYZINormal / PointSurfel keep their curvature at +36 / +44 with rgb or intensity at +32.
struct RecordLayout {
  bool xyz_at_0 = false;         // x, y, z FLOAT32 at +0, +4, +8
  bool normal_at_16 = false;     // normal_x, normal_y, normal_z FLOAT32 at +16, +20, +24
  bool curvature_at_32 = false;  // curvature FLOAT32 at +32
  bool other_fields = false;     // anything else (or one of the above somewhere else)
};
template <typename PointT>
inline const RecordLayout& record_layout() {
  static const RecordLayout layout = [] {
    RecordLayout L;
    std::vector<pcl::PCLPointField> fields;
    (void)pcl::getFieldIndex<PointT>("x", fields);
    unsigned xyz = 0, nrm = 0;
    for (const pcl::PCLPointField& f : fields) {
      const bool f32 = f.datatype == pcl::PCLPointField::FLOAT32 && f.count <= 1;
      const auto at = [&](const char* name, std::uint32_t offset) { return f.name == name && f.offset == offset && f32; };
      if (at("x", 0)) xyz |= 1u;
      else if (at("y", 4)) xyz |= 2u;
      else if (at("z", 8)) xyz |= 4u;
      else if (at("normal_x", 16)) nrm |= 1u;
      else if (at("normal_y", 20)) nrm |= 2u;
      else if (at("normal_z", 24)) nrm |= 4u;
      else if (at("curvature", 32)) L.curvature_at_32 = true;
      else L.other_fields = true;
    }
    L.xyz_at_0 = xyz == 7u;
    L.normal_at_16 = nrm == 7u && sizeof(PointT) >= 32;
    if (nrm != 0 && nrm != 7u) L.other_fields = true;
    L.curvature_at_32 = L.curvature_at_32 && sizeof(PointT) >= 36;
    return L;
  }();
  return layout;
}
// large enough for the normal (a necessary condition, for static_asserts); has_normal_fields() is the real test
template <typename PointT> constexpr bool record_fits_normals() { return sizeof(PointT) >= 32; }
template <typename PointT> inline bool has_normal_fields() { return record_layout<PointT>().normal_at_16; }

// ---- search backend ---------------------------------------------------------------------------------
template <typename PointT>
class KdTreeHIP : public pcl::search::KdTree<PointT> {
  using Base = pcl::search::KdTree<PointT>;
 public:
  using Ptr = std::shared_ptr<KdTreeHIP<PointT>>;
  using PointCloudConstPtr = typename Base::PointCloudConstPtr;
  using IndicesConstPtr = pcl::IndicesConstPtr;
  using PointRepresentationConstPtr = typename Base::PointRepresentationConstPtr;

  // the protected constructor leaves the FLANN tree_ unset (kdtree.h:165-167), as KdTreeNanoflann does
  explicit KdTreeHIP(Device::Ptr dev = Device::instance(), bool sorted = true) : Base("KdTreeHIP", sorted), dev_(std::move(dev)) {}
  ~KdTreeHIP() override { if (index_) pclhip_index_destroy(index_); }
  KdTreeHIP(const KdTreeHIP&) = delete;
  KdTreeHIP& operator=(const KdTreeHIP&) = delete;

  // search/include/pcl/search/impl/kdtree.hpp:87-97: always rebuilds, like the reference
  bool setInputCloud(const PointCloudConstPtr& cloud, const IndicesConstPtr& indices = IndicesConstPtr()) override {
    if (index_) { pclhip_index_destroy(index_); index_ = nullptr; }
    ++generation_;  // users that cache device state per index compare this, not the handle (a new index may reuse the address)
    this->input_ = cloud;
    this->indices_ = indices;
    if (!dev_ || !dev_->ok() || !cloud) return false;
    const bool use_idx = indices && !indices->empty();
    // PointNormal-like records bring their normals along in the same upload (point-to-plane targets)
    const RecordLayout& rec = record_layout<PointT>();
    const bool with_normals = rec.normal_at_16 && rec.curvature_at_32;
    normals_from_ = nullptr;
    const bool ok = pclhip_index_build_ex(dev_->get(), cloud->points.data(), sizeof(PointT), cloud->size(),
                                          use_idx ? indices->data() : nullptr, use_idx ? indices->size() : 0,
                                          scaled_ ? scale_ : nullptr, with_normals ? 16 : 0, &index_) == PCLHIP_OK;
    if (ok && with_normals) normals_from_ = cloud.get();
    return ok;
  }
  // the cloud whose own normals the index carries (nullptr: none attached by setInputCloud)
  const void* normalsFrom() const { return normals_from_; }
  // The index is three-dimensional: a representation is honoured when it is x, y, z (or a prefix of them)
  // with per-axis rescale factors (kdtree.h:110 -> KdTreeFLANN::setPointRepresentation); anything else is
  // refused loudly -- searches would silently mean something different.
  void setPointRepresentation(const PointRepresentationConstPtr& rep) override {
    rep_ = rep;
    scaled_ = false;
    if (!rep) return;
    const int d = rep->getNumberOfDimensions();
    float probe[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    bool ok = d >= 1 && d <= 3;
    for (int a = 0; a < 3 && ok; ++a) {
      PointT p;
      p.x = a == 0 ? 1.0f : 0.0f; p.y = a == 1 ? 1.0f : 0.0f; p.z = a == 2 ? 1.0f : 0.0f;
      std::vector<float> out(size_t(d), 0.0f);
      rep->vectorize(p, out);
      for (int j = 0; j < d; ++j) probe[a][j] = out[size_t(j)];
    }
    for (int a = 0; a < 3 && ok; ++a)
      for (int j = 0; j < 3; ++j)
        if (j != a && probe[a][j] != 0.0f) ok = false;  // not a diagonal map of (x, y, z)
    if (!ok) {
      unsupported_ = true;
      return;
    }
    unsupported_ = false;
    for (int a = 0; a < 3; ++a) scale_[a] = a < d ? probe[a][a] : 0.0f;
    scaled_ = !(scale_[0] == 1.0f && scale_[1] == 1.0f && scale_[2] == 1.0f);
    if (this->input_) setInputCloud(this->input_, this->indices_);
  }
  PointRepresentationConstPtr getPointRepresentation() const override { return rep_; }
  bool representationSupported() const { return !unsupported_; }
  void setEpsilon(float eps) override { eps_ = eps; }  // the search is exact: an error bound is trivially met
  float getEpsilon() const override { return eps_; }

  // kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp:234-274 (one launch per call: API parity, not the fast path).
  // The `const` per-point searches may be called from several threads at once, as PCL's OpenMP loops do
  // (registration/.../impl/correspondence_estimation.hpp:163-175 with setNumberOfThreads, features/.../normal_3d_omp.hpp
  // :76-81, search/.../impl/search.hpp:164-190): pclhip_knn / pclhip_radius_search serialise on the context (pclhip.h),
  // and nothing of this object is written by them.
  int nearestKSearch(const PointT& p, int k, pcl::Indices& k_indices, std::vector<float>& k_sqr_distances) const override {
    k_indices.clear();
    k_sqr_distances.clear();
    if (!index_ || unsupported_ || k < 1) return 0;
    const std::uint64_t n = pclhip_index_size(index_);
    if (std::uint64_t(k) > n) k = int(n);
    if (k == 0) return 0;
    k_indices.resize(size_t(k));
    k_sqr_distances.resize(size_t(k));
    if (pclhip_knn(index_, &p, sizeof(PointT), 1, k, k_indices.data(), k_sqr_distances.data()) != PCLHIP_OK) return 0;
    int found = 0;
    while (found < k && k_indices[size_t(found)] >= 0) ++found;
    k_indices.resize(size_t(found));
    k_sqr_distances.resize(size_t(found));
    return found;
  }
  // the batch overload (search.h:216-219): ONE launch for the whole cloud
  void nearestKSearch(const pcl::PointCloud<PointT>& cloud, const pcl::Indices& indices, int k,
                      std::vector<pcl::Indices>& k_indices, std::vector<std::vector<float>>& k_sqr_distances) const override {
    k_indices.clear();
    k_sqr_distances.clear();
    if (!index_ || unsupported_ || k < 1) return;
    std::vector<PointT> gathered;
    const PointT* q = cloud.points.data();
    std::size_t nq = cloud.size();
    if (!indices.empty()) {
      gathered.reserve(indices.size());
      for (pcl::index_t i : indices) gathered.push_back(cloud[size_t(i)]);
      q = gathered.data();
      nq = gathered.size();
    }
    const std::uint64_t n = pclhip_index_size(index_);
    const int kk = std::uint64_t(k) > n ? int(n) : k;
    k_indices.assign(nq, pcl::Indices());
    k_sqr_distances.assign(nq, std::vector<float>());
    if (kk == 0 || nq == 0) return;
    pcl::Indices flat_i(nq * size_t(kk));
    std::vector<float> flat_d(nq * size_t(kk));
    if (pclhip_knn(index_, q, sizeof(PointT), nq, kk, flat_i.data(), flat_d.data()) != PCLHIP_OK) return;
    for (std::size_t i = 0; i < nq; ++i) {
      int found = 0;
      while (found < kk && flat_i[i * size_t(kk) + size_t(found)] >= 0) ++found;
      k_indices[i].assign(flat_i.begin() + long(i * size_t(kk)), flat_i.begin() + long(i * size_t(kk)) + found);
      k_sqr_distances[i].assign(flat_d.begin() + long(i * size_t(kk)), flat_d.begin() + long(i * size_t(kk)) + found);
    }
  }
  // kdtree_flann.hpp:372-414: squared distance < radius^2, ascending; max_nn == 0: unlimited
  int radiusSearch(const PointT& p, double radius, pcl::Indices& k_indices, std::vector<float>& k_sqr_distances,
                   unsigned int max_nn = 0) const override {
    k_indices.clear();
    k_sqr_distances.clear();
    if (!index_ || unsupported_) return 0;
    std::uint64_t off[2] = {0, 0}, total = 0;
    const pclhip_status st = pclhip_radius_search(index_, &p, sizeof(PointT), 1, radius, max_nn, off, nullptr, nullptr, 0, &total);
    if ((st != PCLHIP_OK && st != PCLHIP_ERR_OVERFLOW) || total == 0) return 0;
    k_indices.resize(size_t(total));
    k_sqr_distances.resize(size_t(total));
    if (pclhip_radius_search(index_, &p, sizeof(PointT), 1, radius, max_nn, off, k_indices.data(), k_sqr_distances.data(),
                             total, &total) != PCLHIP_OK) {
      k_indices.clear();
      k_sqr_distances.clear();
      return 0;
    }
    return int(total);
  }
  // the batch overload (search.h:349-355; the default, impl/search.hpp:164-190, is an OpenMP loop over the per-point
  // virtual): ONE pclhip_radius_search for the whole cloud, the lists handed out from its CSR result
  void radiusSearch(const pcl::PointCloud<PointT>& cloud, const pcl::Indices& indices, double radius,
                    std::vector<pcl::Indices>& k_indices, std::vector<std::vector<float>>& k_sqr_distances,
                    unsigned int max_nn = 0) const override {
    k_indices.clear();
    k_sqr_distances.clear();
    if (!index_ || unsupported_) return;
    std::vector<PointT> gathered;
    const PointT* q = cloud.points.data();
    std::size_t nq = cloud.size();
    if (!indices.empty()) {
      gathered.reserve(indices.size());
      for (pcl::index_t i : indices) gathered.push_back(cloud[size_t(i)]);
      q = gathered.data();
      nq = gathered.size();
    }
    k_indices.assign(nq, pcl::Indices());
    k_sqr_distances.assign(nq, std::vector<float>());
    if (nq == 0) return;
    std::vector<std::uint64_t> off(nq + 1, 0);
    std::uint64_t total = 0;
    const pclhip_status st = pclhip_radius_search(index_, q, sizeof(PointT), nq, radius, max_nn, off.data(), nullptr, nullptr, 0, &total);
    if ((st != PCLHIP_OK && st != PCLHIP_ERR_OVERFLOW) || total == 0) return;
    pcl::Indices flat_i(size_t(total), 0);
    std::vector<float> flat_d(size_t(total), 0.0f);
    if (pclhip_radius_search(index_, q, sizeof(PointT), nq, radius, max_nn, off.data(), flat_i.data(), flat_d.data(), total,
                             &total) != PCLHIP_OK)
      return;
    for (std::size_t i = 0; i < nq; ++i) {
      k_indices[i].assign(flat_i.begin() + long(off[i]), flat_i.begin() + long(off[i + 1]));
      k_sqr_distances[i].assign(flat_d.begin() + long(off[i]), flat_d.begin() + long(off[i + 1]));
    }
  }
  pclhip_index* handle() const { return index_; }
  std::uint64_t generation() const { return generation_; }  // bumped by every setInputCloud
  const Device::Ptr& device() const { return dev_; }

 private:
  Device::Ptr dev_;
  pclhip_index* index_ = nullptr;
  std::uint64_t generation_ = 0;
  const void* normals_from_ = nullptr;
  PointRepresentationConstPtr rep_;
  float scale_[3] = {1, 1, 1};
  bool scaled_ = false, unsupported_ = false;
  float eps_ = 0.0f;
};

// ---- the batch correspondence entry -------------------------------------------------------------------
template <typename PointSource, typename PointTarget, typename Scalar = float>
class CorrespondenceEstimationHIP : public pcl::registration::CorrespondenceEstimationBase<PointSource, PointTarget, Scalar> {
  using Base = pcl::registration::CorrespondenceEstimationBase<PointSource, PointTarget, Scalar>;
 public:
  using Ptr = std::shared_ptr<CorrespondenceEstimationHIP<PointSource, PointTarget, Scalar>>;
  CorrespondenceEstimationHIP() { this->corr_name_ = "CorrespondenceEstimationHIP"; }
  ~CorrespondenceEstimationHIP() override { if (icp_) pclhip_icp_destroy(icp_); }
  CorrespondenceEstimationHIP(const CorrespondenceEstimationHIP& o) : Base(o) { this->corr_name_ = "CorrespondenceEstimationHIP"; }

  // impl/correspondence_estimation.hpp:145-218 for the whole cloud in one launch
  void determineCorrespondences(pcl::Correspondences& correspondences,
                                double max_distance = std::numeric_limits<double>::max()) override {
    run(correspondences, max_distance, false);
  }
  // :220-311
  void determineReciprocalCorrespondences(pcl::Correspondences& correspondences,
                                          double max_distance = std::numeric_limits<double>::max()) override {
    run(correspondences, max_distance, true);
  }
  typename Base::Ptr clone() const override {
    return typename Base::Ptr(new CorrespondenceEstimationHIP<PointSource, PointTarget, Scalar>(*this));
  }

 private:
  void run(pcl::Correspondences& out, double max_distance, bool reciprocal) {
    out.clear();
    if (!this->initCompute()) return;  // PCL's own bookkeeping: (re)builds the target tree when it changed
    auto* tree = dynamic_cast<KdTreeHIP<PointTarget>*>(this->tree_.get());
    if (tree == nullptr || tree->handle() == nullptr) { this->deinitCompute(); return; }  // needs the HIP search backend
    if (icp_ && (icp_target_ != tree->handle() || icp_generation_ != tree->generation())) { pclhip_icp_destroy(icp_); icp_ = nullptr; }
    if (!icp_) {
      if (pclhip_icp_create(tree->handle(), &icp_) != PCLHIP_OK) { this->deinitCompute(); return; }
      icp_target_ = tree->handle();
      icp_generation_ = tree->generation();
    }
    // a user-set index list (any size, any order, duplicates allowed) -- not the one PCLBase::initCompute fills in
    const bool subset = this->indices_ && !this->fake_indices_;
    static const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double sums[PCLHIP_ICP_NSUMS];
    // the squared distances of ALL pairs up to max_distance are wanted: pass the distance itself
    const double md = max_distance < 1e150 ? max_distance : 1e150;
    if (pclhip_icp_set_source_indexed(icp_, this->input_->points.data(), sizeof(PointSource), this->input_->size(),
                                      subset ? this->indices_->data() : nullptr, subset ? this->indices_->size() : 0) == PCLHIP_OK &&
        pclhip_icp_set_reciprocal(icp_, reciprocal ? 1 : 0) == PCLHIP_OK &&
        pclhip_icp_iterate(icp_, I, md, PCLHIP_ICP_POINT_TO_POINT, sums) == PCLHIP_OK) {
      const std::size_t n = this->input_->size();
      if (!subset && sizeof(pcl::Correspondence) == 12) {
        // the whole cloud, ascending by index_query (the order of the reference's loop over all points): the device
        // compacts the kept pairs into pcl::Correspondence records and the vector receives them in one copy -- sized for
        // every query first, shrunk to the valid ones afterwards, as impl/correspondence_estimation.hpp:160-216 does
        out.resize(n);
        std::uint64_t cnt = 0;
        if (pclhip_icp_fetch_correspondence_records(icp_, out.data(), n, &cnt) == PCLHIP_OK) {
          out.resize(std::size_t(cnt));
          this->deinitCompute();
          return;
        }
        out.clear();
      }
      pcl::Indices q(n), m(n);
      std::vector<float> d(n);
      std::uint64_t cnt = 0;
      if (pclhip_icp_fetch_correspondences(icp_, q.data(), m.data(), d.data(), &cnt) == PCLHIP_OK) {
        if (!subset) {  // ascending by index_query: the order of the reference's loop over all points
          out.reserve(size_t(cnt));
          for (std::uint64_t i = 0; i < cnt; ++i) out.emplace_back(q[size_t(i)], m[size_t(i)], d[size_t(i)]);
        } else {
          // the reference emits one correspondence per ENTRY of indices_, in that order (impl/correspondence_estimation.hpp
          // :160-216): the device returns every distinct query once, ascending -- map back through the list
          std::vector<std::int64_t> at(n, -1);
          for (std::uint64_t i = 0; i < cnt; ++i) at[size_t(q[size_t(i)])] = std::int64_t(i);
          out.reserve(this->indices_->size());
          for (const pcl::index_t idx : *this->indices_) {
            const std::int64_t i = (idx >= 0 && size_t(idx) < n) ? at[size_t(idx)] : -1;
            if (i >= 0) out.emplace_back(idx, m[size_t(i)], d[size_t(i)]);
          }
        }
      }
    }
    this->deinitCompute();
  }
  pclhip_icp* icp_ = nullptr;
  pclhip_index* icp_target_ = nullptr;
  std::uint64_t icp_generation_ = 0;
};

// ---- the whole loop on the device ------------------------------------------------------------------------
// Shared body of the two ICP subclasses.  `Base` is pcl::IterativeClosestPoint<S,T,Scalar> or
// pcl::IterativeClosestPointWithNormals<S,T,Scalar>.  Scalar (registration.h:56, icp.h:97,339: float by default, double
// for callers that keep their poses in double) is the type of the 4x4 matrices at the boundary: the points are float
// either way and the reference moves them with the matrix cast to float (impl/icp.hpp:54-55); the device computes what it
// computes for float -- float products, fp64 sums, fp64 solve -- and a double instantiation receives those matrices
// widened.  PCL's own double instantiation keeps final = T_k * final in double: the two agree to float rounding of the
// 4x4 (the 1e-5 Frobenius contract of SURVEY.md section 8), not bit for bit.
template <typename Base, typename PointSource, typename PointTarget, typename Scalar = float>
class RegistrationHIP : public Base {
 public:
  using PointCloudSource = pcl::PointCloud<PointSource>;
  using Matrix4 = typename Base::Matrix4;
  ~RegistrationHIP() override { if (icp_) pclhip_icp_destroy(icp_); }
  // Why the last align() ran where it ran: "" = device; otherwise the reason it deferred to PCL's CPU loop.
  const std::string& deferredReason() const { return deferred_; }
  // Host wall time of the parts of the last computeTransformation (milliseconds): what the boundary costs a PCL user
  // next to the iterations themselves.  upload: target normals + source records to the device (0 when they were
  // resident); loop: pclhip_icp_align (source ordering included when the source was new); output: the moved cloud.
  struct Timings { double upload_ms = 0, loop_ms = 0, output_ms = 0; };
  const Timings& lastTimings() const { return timings_; }
  int iterations() const { return this->nr_iterations_; }  // of the last align() (PCL keeps nr_iterations_ protected)
  // Registration::getFitnessScore (registration.h:450-452, impl/registration.hpp:132-168) on the device, after an align().
  // The reference's is NOT virtual: this one HIDES it -- called on the HIP class it scores on the device in one launch; a
  // caller holding a pcl::Registration* gets PCL's own loop, which reaches the same index point by point through
  // KdTreeHIP::nearestKSearch (same value, one launch per point).  The device holds the source as the last align()
  // uploaded it -- the whole cloud, or the user's index subset --: when the reference would score the other set
  // (use_indices, :141-144), PCL's loop runs.
  double getFitnessScore(double max_range = std::numeric_limits<double>::max(), bool use_indices = false) {
    const bool subset = this->indices_ && this->input_ && this->indices_->size() != this->input_->size();
    if (!icp_ || source_uploaded_ != this->input_.get() || (subset && use_indices != source_subset_) || (!subset && source_subset_))
      return Base::getFitnessScore(max_range, use_indices);
    float T[16];
    to_rows(this->final_transformation_, T);
    double score = std::numeric_limits<double>::max();
    if (pclhip_icp_fitness_score(icp_, T, max_range, &score, nullptr) != PCLHIP_OK) return Base::getFitnessScore(max_range, use_indices);
    return score;
  }

 protected:
  enum Kind { SVD = PCLHIP_ICP_POINT_TO_POINT, LLS = PCLHIP_ICP_POINT_TO_PLANE, SYM = PCLHIP_ICP_SYMMETRIC, FOREIGN = -1 };
  virtual bool enforceSameDirectionNormals() const { return true; }

  static void to_rows(const Matrix4& M, float* T) {
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) T[4 * r + c] = float(M(r, c));
  }
  static void from_rows(const float* T, Matrix4& M) {
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) M(r, c) = Scalar(T[4 * r + c]);
  }
  Kind estimatorKind() const {
    using namespace pcl::registration;
    const auto* te = this->transformation_estimation_.get();
    if (dynamic_cast<const TransformationEstimationSymmetricPointToPlaneLLS<PointSource, PointTarget, Scalar>*>(te)) return SYM;
    if (dynamic_cast<const TransformationEstimationPointToPlaneLLS<PointSource, PointTarget, Scalar>*>(te)) return LLS;
    if (dynamic_cast<const TransformationEstimationSVD<PointSource, PointTarget, Scalar>*>(te)) return SVD;
    return FOREIGN;
  }
  // the rejectors the device chain knows, recognised by class name (registration.h:518-547)
  bool mapRejectors(std::vector<pclhip_rejector>& out) const {
    using namespace pcl::registration;
    for (const auto& r : this->correspondence_rejectors_) {
      pclhip_rejector d{PCLHIP_REJ_DISTANCE, 0.0, 0, 0};
      const std::string& name = r->getClassName();
      if (name == "CorrespondenceRejectorDistance") {
        d.kind = PCLHIP_REJ_DISTANCE;
        d.param = static_cast<const CorrespondenceRejectorDistance*>(r.get())->getMaximumDistance();
      } else if (name == "CorrespondenceRejectorMedianDistance") {
        d.kind = PCLHIP_REJ_MEDIAN_DISTANCE;
        d.param = static_cast<const CorrespondenceRejectorMedianDistance*>(r.get())->getMedianFactor();
      } else if (name == "CorrespondenceRejectorOneToOne") {
        d.kind = PCLHIP_REJ_ONE_TO_ONE;
      } else if (name == "CorrespondenceRejectorTrimmed") {
        const auto* t = static_cast<const CorrespondenceRejectorTrimmed*>(r.get());
        d.kind = PCLHIP_REJ_TRIMMED;
        d.param = t->getOverlapRatio();
        d.min_correspondences = t->getMinCorrespondences();
      } else {
        return false;
      }
      out.push_back(d);
    }
    return true;
  }

  // impl/icp.hpp:113-268 on the device.  Anything this path cannot express exactly -- a foreign search
  // backend, estimator, correspondence estimator or rejector -- defers to PCL's own loop, so behaviour
  // never changes silently.
  void computeTransformation(PointCloudSource& output, const Matrix4& guess) override {
    using namespace pcl::registration;
    deferred_.clear();
    auto* tree = dynamic_cast<KdTreeHIP<PointTarget>*>(this->tree_.get());
    const Kind kind = estimatorKind();
    std::vector<pclhip_rejector> rej;
    const auto* ce = this->correspondence_estimation_.get();
    const bool own_ce = ce == nullptr || dynamic_cast<const CorrespondenceEstimationHIP<PointSource, PointTarget, Scalar>*>(ce) ||
                        dynamic_cast<const CorrespondenceEstimation<PointSource, PointTarget, Scalar>*>(ce);
    if (!record_layout<PointSource>().xyz_at_0 || !record_layout<PointTarget>().xyz_at_0)
      deferred_ = "a point type without (x, y, z) floats at offset 0";
    else if (tree == nullptr || tree->handle() == nullptr) deferred_ = "the target search method is not a KdTreeHIP";
    else if (!tree->representationSupported()) deferred_ = "the point representation is not a rescaled (x, y, z)";
    else if (kind == FOREIGN) deferred_ = "foreign TransformationEstimation";
    else if (!own_ce) deferred_ = "foreign CorrespondenceEstimation";
    else if (!mapRejectors(rej)) deferred_ = "foreign CorrespondenceRejector";
    else if (kind != SVD && !has_normal_fields<PointTarget>()) deferred_ = "point-to-plane needs a target type with normals";
    else if (kind == SYM && !has_normal_fields<PointSource>()) deferred_ = "the symmetric objective needs a source type with normals";
    if (!deferred_.empty()) {
      Base::computeTransformation(output, guess);
      return;
    }
    // device state is cached per BUILD of the target index (KdTreeHIP::generation), not per handle address: a rebuilt
    // index may land on the freed address, and it has neither the normals nor this registration attached
    using Clock = std::chrono::steady_clock;
    const auto ms = [](Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const Clock::time_point t_start = Clock::now();
    timings_ = Timings();
    if (icp_ && (icp_target_ != tree->handle() || icp_generation_ != tree->generation())) { pclhip_icp_destroy(icp_); icp_ = nullptr; }
    if (!icp_) {
      if (pclhip_icp_create(tree->handle(), &icp_) != PCLHIP_OK) return;
      icp_target_ = tree->handle();
      icp_generation_ = tree->generation();
      source_uploaded_ = nullptr;
      target_normals_of_ = nullptr;
    }
    // KdTreeHIP::setInputCloud attaches a PointNormal cloud's normals from its own upload.  They are the cloud's normals
    // of THAT moment: trusted only when Registration::initCompute built the tree itself (registration.hpp:73-101 rebuilds
    // inside align() unless force_no_recompute_).  A tree the user built beforehand (setSearchMethodTarget(tree, true))
    // may predate the normals -- NormalEstimation writing into the same cloud afterwards --, so its first alignment
    // uploads them, as PCL's CPU path would read them now.
    if (kind != SVD && !this->force_no_recompute_ && tree->normalsFrom() == static_cast<const void*>(this->target_.get()))
      target_normals_of_ = this->target_.get();
    if (kind != SVD && target_normals_of_ != this->target_.get()) {  // pcl::PointNormal: normals at +16
      const char* base = reinterpret_cast<const char*>(this->target_->points.data());
      if (pclhip_index_set_normals(tree->handle(), base + 16, sizeof(PointTarget)) != PCLHIP_OK) return;
      target_normals_of_ = this->target_.get();
    }
    // a user-set index list: PCLBase::setIndices does not flag the source as updated, so a subset is uploaded every time
    const bool subset = this->indices_ && !this->fake_indices_;
    if (source_uploaded_ != this->input_.get() || this->source_cloud_updated_ || subset || source_subset_) {
      if (pclhip_icp_set_source_indexed(icp_, this->input_->points.data(), sizeof(PointSource), this->input_->size(),
                                        subset ? this->indices_->data() : nullptr, subset ? this->indices_->size() : 0) != PCLHIP_OK)
        return;
      if (kind == SYM) {
        const char* base = reinterpret_cast<const char*>(this->input_->points.data());
        if (pclhip_icp_set_source_normals(icp_, base + 16, sizeof(PointSource)) != PCLHIP_OK) return;
      }
      source_uploaded_ = this->input_.get();
      source_subset_ = subset;
      this->source_cloud_updated_ = false;
    }
    if (pclhip_icp_set_rejectors(icp_, rej.data(), int(rej.size())) != PCLHIP_OK) return;
    if (pclhip_icp_set_reciprocal(icp_, this->use_reciprocal_correspondence_ ? 1 : 0) != PCLHIP_OK) return;
    if (pclhip_icp_set_enforce_same_direction_normals(icp_, enforceSameDirectionNormals() ? 1 : 0) != PCLHIP_OK) return;

    pclhip_icp_params p;
    pclhip_icp_params_default(&p);
    p.mode = int(kind);
    p.max_iterations = this->max_iterations_;
    p.max_correspondence_distance = this->corr_dist_threshold_;
    p.transformation_epsilon = this->transformation_epsilon_;
    p.transformation_rotation_epsilon = this->transformation_rotation_epsilon_;
    p.euclidean_fitness_epsilon = this->euclidean_fitness_epsilon_;
    p.min_number_correspondences = int(this->min_number_correspondences_);
    p.failure_after_max_iterations = this->convergence_criteria_->getFailureAfterMaximumIterations() ? 1 : 0;
    p.max_iterations_similar_transforms = this->convergence_criteria_->getMaximumIterationsSimilarTransforms();
    p.mse_threshold_absolute = this->convergence_criteria_->getAbsoluteMSE();
    float g[16];
    to_rows(guess, g);
    pclhip_icp_result r;
    const Clock::time_point t_loop = Clock::now();
    timings_.upload_ms = ms(t_start, t_loop);
    if (pclhip_icp_align(icp_, &p, g, &r) != PCLHIP_OK) return;
    const Clock::time_point t_out = Clock::now();
    timings_.loop_ms = ms(t_loop, t_out);
    from_rows(r.final_transformation, this->final_transformation_);
    from_rows(r.last_transformation, this->transformation_);
    this->nr_iterations_ = r.nr_iterations;
    this->converged_ = r.converged != 0;
    using Criteria = DefaultConvergenceCriteria<Scalar>;
    this->convergence_criteria_->setConvergenceState(static_cast<typename Criteria::ConvergenceState>(r.convergence_state));
    // impl/icp.hpp:264-267: the whole input cloud, moved by the final transformation.  (pcl::Registration::align hands
    // computeTransformation a copy of the input already -- registration.hpp:190-205 -- so the records are only copied
    // again when the caller's cloud is not that copy.)
    if (&output != this->input_.get() && (output.size() != this->input_->size() || output.points.data() == nullptr))
      output = *this->input_;
    else if (&output != this->input_.get()) {
      output.header = this->input_->header;
      output.width = this->input_->width;
      output.height = this->input_->height;
      output.is_dense = this->input_->is_dense;
    }
    pclhip_icp_transform_source(icp_, r.final_transformation, kind == SVD ? 0 : 1, this->input_->points.data(),
                                output.points.data(), sizeof(PointSource), output.size(),
                                (kind != SVD && has_normal_fields<PointSource>()) ? 16 : 0);
    timings_.output_ms = ms(t_out, Clock::now());
  }

 private:
  pclhip_icp* icp_ = nullptr;
  pclhip_index* icp_target_ = nullptr;
  std::uint64_t icp_generation_ = 0;
  const void* source_uploaded_ = nullptr;
  const void* target_normals_of_ = nullptr;
  bool source_subset_ = false;
  std::string deferred_;
  Timings timings_;
};

template <typename PointSource, typename PointTarget, typename Scalar = float>
class IterativeClosestPointHIP
    : public RegistrationHIP<pcl::IterativeClosestPoint<PointSource, PointTarget, Scalar>, PointSource, PointTarget, Scalar> {
 public:
  using Ptr = std::shared_ptr<IterativeClosestPointHIP<PointSource, PointTarget, Scalar>>;
  explicit IterativeClosestPointHIP(Device::Ptr dev = Device::instance()) {
    this->reg_name_ = "IterativeClosestPointHIP";
    this->setSearchMethodTarget(std::make_shared<KdTreeHIP<PointTarget>>(dev));
    this->setCorrespondenceEstimation(std::make_shared<CorrespondenceEstimationHIP<PointSource, PointTarget, Scalar>>());
  }
};

template <typename PointSource, typename PointTarget, typename Scalar = float>
class IterativeClosestPointWithNormalsHIP
    : public RegistrationHIP<pcl::IterativeClosestPointWithNormals<PointSource, PointTarget, Scalar>, PointSource, PointTarget, Scalar> {
 public:
  using Ptr = std::shared_ptr<IterativeClosestPointWithNormalsHIP<PointSource, PointTarget, Scalar>>;
  explicit IterativeClosestPointWithNormalsHIP(Device::Ptr dev = Device::instance()) {
    this->reg_name_ = "IterativeClosestPointWithNormalsHIP";
    this->setSearchMethodTarget(std::make_shared<KdTreeHIP<PointTarget>>(dev));
    this->setCorrespondenceEstimation(std::make_shared<CorrespondenceEstimationHIP<PointSource, PointTarget, Scalar>>());
  }
 protected:
  bool enforceSameDirectionNormals() const override { return this->getEnforceSameDirectionNormals(); }
};

// ---- NormalEstimation ---------------------------------------------------------------------------------
// Feature::compute (impl/feature.hpp:195-229) has already run initCompute when computeFeature is called: the tree
// indexes surface_, indices_ holds the queries, exactly one of k_ / search_radius_ is set.  The tree must be a
// KdTreeHIP (Feature::setSearchMethod); with any other search object the CPU implementation of the base runs.
template <typename PointInT, typename PointOutT>
class NormalEstimationHIP : public pcl::NormalEstimation<PointInT, PointOutT> {
  using Base = pcl::NormalEstimation<PointInT, PointOutT>;
 public:
  using Ptr = std::shared_ptr<NormalEstimationHIP<PointInT, PointOutT>>;
  using PointCloudOut = typename Base::PointCloudOut;
  explicit NormalEstimationHIP(Device::Ptr dev = Device::instance()) {
    this->feature_name_ = "NormalEstimationHIP";
    this->setSearchMethod(std::make_shared<KdTreeHIP<PointInT>>(std::move(dev)));
  }
  // why the last compute() ran PCL's own loop instead of the device ("" = it ran on the device)
  const std::string& deferredReason() const { return deferred_; }
 protected:
  void computeFeature(PointCloudOut& output) override {  // impl/normal_3d.hpp:48-95
    auto* dev_tree = dynamic_cast<KdTreeHIP<PointInT>*>(this->tree_.get());
    if (dev_tree == nullptr || dev_tree->handle() == nullptr || !dev_tree->representationSupported()) {
      deferred_ = "the search method is not a KdTreeHIP";
      Base::computeFeature(output);
      return;
    }
    deferred_.clear();
    const std::size_t m = this->indices_->size();
    std::uint64_t nan = 0;
    const float vp[3] = {this->vpx_, this->vpy_, this->vpz_};
    const bool all = this->fake_indices_;
    pclhip_status st;
    if constexpr (std::is_same<PointOutT, pcl::Normal>::value) {
      // every point of the cloud itself, into pcl::Normal records: the device hands back whole records (zeros, the normal at
      // +0, the curvature at +16 -- what Feature::compute's resize + computeFeature leave), one linear copy straight into
      // the output cloud instead of a 160 MB staging vector (fresh pages on every call) and a 10M-iteration unpacking loop
      if (this->fake_surface_ && all && output.size() == m && m == this->input_->size()) {
        st = pclhip_normals_records(dev_tree->handle(), this->k_, this->k_ != 0 ? 0.0 : this->search_radius_, vp,
                                    output.points.data(), sizeof(pcl::Normal), 0, 16, &nan);
        if (st == PCLHIP_OK) {
          if (nan != 0) output.is_dense = false;
          return;
        }
      }
    }
    std::vector<float>& tmp = staging_;  // kept between calls: a fresh 160 MB vector costs 40 ms of page faults every time
    tmp.resize(m * 4);
    if (this->fake_surface_ && all) {  // surface == input, every point: the fused kernel, normals kept in the index
      st = (this->k_ != 0) ? pclhip_normals(dev_tree->handle(), this->k_, vp, tmp.data(), 16, &nan)
                           : pclhip_normals_radius(dev_tree->handle(), this->search_radius_, vp, tmp.data(), 16, &nan);
    } else {
      st = pclhip_normals_at(dev_tree->handle(), this->input_->points.data(), sizeof(PointInT), this->input_->size(),
                             all ? nullptr : this->indices_->data(), all ? 0 : m, this->k_, this->k_ != 0 ? 0.0 : this->search_radius_,
                             vp, tmp.data(), 16, &nan);
    }
    if (st != PCLHIP_OK) {
      const float qnan = std::numeric_limits<float>::quiet_NaN();
      for (auto& v : tmp) v = qnan;
      nan = m;
    }
    for (std::size_t i = 0; i < m; ++i) {  // normal_3d.hpp:60-66,79-91
      output[i].normal[0] = tmp[4 * i]; output[i].normal[1] = tmp[4 * i + 1]; output[i].normal[2] = tmp[4 * i + 2];
      output[i].curvature = tmp[4 * i + 3];
    }
    if (nan != 0) output.is_dense = false;
  }
 private:
  std::string deferred_;
  std::vector<float> staging_;
};

// ---- VoxelGrid ----------------------------------------------------------------------------------------
template <typename PointT>
class VoxelGridHIP : public pcl::VoxelGrid<PointT> {
  using Base = pcl::VoxelGrid<PointT>;
 public:
  using Ptr = std::shared_ptr<VoxelGridHIP<PointT>>;
  using PointCloud = typename pcl::Filter<PointT>::PointCloud;
  explicit VoxelGridHIP(Device::Ptr dev = Device::instance()) : dev_(std::move(dev)) { this->filter_name_ = "VoxelGridHIP"; }
  const std::string& deferredReason() const { return deferred_; }
  // Why the next filter() call would run pcl::VoxelGrid's own applyFilter instead of the device path ("" = it would
  // not).  With setDownsampleAllData(true) the reference averages EVERY field of the point type
  // (impl/voxel_grid.hpp:760-800, CentroidPoint); the device path averages the coordinates and PointNormal's normal and
  // curvature, so any other field list goes back to the reference.  Coordinates only: any type with xyz at +0.
  std::string whyDeferred() const {
    const RecordLayout& rec = record_layout<PointT>();
    if (!dev_ || !dev_->ok()) return "no device";
    if (this->indices_ && !this->fake_indices_) return "an index subset of the input";  // the device path filters whole clouds
    if (!rec.xyz_at_0) return "a point type without (x, y, z) floats at offset 0";
    if (this->downsample_all_data_ && (rec.other_fields || rec.normal_at_16 != rec.curvature_at_32))
      return "setDownsampleAllData(true) on a point type with fields beyond PointXYZ's or PointNormal's";
    return "";
  }
 protected:
  void applyFilter(PointCloud& output) override {  // impl/voxel_grid.hpp:597-814
    deferred_.clear();
    deferred_ = whyDeferred();
    if (!deferred_.empty()) {
      Base::applyFilter(output);
      return;
    }
    // the extras the device averages along with the coordinates: PointNormal's, and only those
    const RecordLayout& rec = record_layout<PointT>();
    const std::size_t extras_at = (this->downsample_all_data_ && rec.normal_at_16 && rec.curvature_at_32) ? 16 : 0;
    output.height = 1;
    output.is_dense = true;
    int limits = 0;
    if (!this->filter_field_name_.empty()) {  // :664-673 -> the field's place inside PointT
      std::vector<pcl::PCLPointField> fields;
      const int idx = pcl::getFieldIndex<PointT>(this->filter_field_name_, fields);
      if (idx < 0 || fields[std::size_t(idx)].offset % 4 != 0) {  // "[applyFilter] Invalid filter field name": the reference stops too
        output.width = 0;
        output.points.clear();
        return;
      }
      limits = PCLHIP_VOXELGRID_LIMITS(int(fields[std::size_t(idx)].offset / 4), this->filter_limit_negative_);
    }
    const float leaf[3] = {this->leaf_size_[0], this->leaf_size_[1], this->leaf_size_[2]};
    const auto& in = *this->input_;
    pclhip_voxelgrid_dims d{};
    this->leaf_layout_.clear();
    std::vector<std::int32_t> layout;
    if (this->save_leaf_layout_ && !in.points.empty() &&
        pclhip_voxelgrid_grid(dev_->get(), in.points.data(), sizeof(PointT), in.size(), leaf, limits, this->filter_limit_min_,
                              this->filter_limit_max_, &d) == PCLHIP_OK)
      layout.assign(std::size_t(d.div_b[0]) * std::size_t(d.div_b[1]) * std::size_t(d.div_b[2]), -1);
    std::vector<PointT> out(in.size());
    std::uint64_t n = 0;
    const pclhip_status st = pclhip_voxelgrid_ex2(dev_->get(), in.points.data(), sizeof(PointT), in.size(), leaf,
                                                  this->min_points_per_voxel_, limits, this->filter_limit_min_,
                                                  this->filter_limit_max_, this->downsample_all_data_ ? 1 : 0,
                                                  extras_at, out.data(), sizeof(PointT), &n,
                                                  layout.empty() ? nullptr : layout.data(), layout.size(), &d);
    if (st == PCLHIP_ERR_OVERFLOW) {  // :620-629: "Leaf size is too small ... Integer indices would overflow"
      output = in;
      return;
    }
    if (st != PCLHIP_OK) {
      output.width = 0;
      output.points.clear();
      return;
    }
    out.resize(std::size_t(n));
    output.points.swap(out);
    output.width = std::uint32_t(n);
    for (int a = 0; a < 3; ++a) {  // :632-647: the grid the getters report
      this->min_b_[a] = d.min_b[a]; this->max_b_[a] = d.max_b[a]; this->div_b_[a] = d.div_b[a]; this->divb_mul_[a] = d.divb_mul[a];
    }
    this->min_b_[3] = this->max_b_[3] = 0; this->div_b_[3] = 1; this->divb_mul_[3] = 0;
    this->leaf_layout_.assign(layout.begin(), layout.end());
  }
 private:
  Device::Ptr dev_;
  std::string deferred_;
};

// ---- transformation estimators on explicit pairs ---------------------------------------------------------
// MODE: PCLHIP_ICP_POINT_TO_POINT (TransformationEstimationSVD), PCLHIP_ICP_POINT_TO_PLANE (...PointToPlaneLLS),
// PCLHIP_ICP_SYMMETRIC (...SymmetricPointToPlaneLLS); normals are the point types' own fields at +16.
template <typename PointSource, typename PointTarget, int MODE, typename Scalar = float>
class TransformationEstimationHIP : public pcl::registration::TransformationEstimation<PointSource, PointTarget, Scalar> {
  using Base = pcl::registration::TransformationEstimation<PointSource, PointTarget, Scalar>;
 public:
  using Matrix4 = typename Base::Matrix4;
  explicit TransformationEstimationHIP(Device::Ptr dev = Device::instance()) : dev_(std::move(dev)) {}
  void setEnforceSameDirectionNormals(bool on) { enforce_ = on; }  // symmetric objective (..._symmetric_point_to_plane_lls.h:123)
  bool getEnforceSameDirectionNormals() const { return enforce_; }
  void estimateRigidTransformation(const pcl::PointCloud<PointSource>& src, const pcl::PointCloud<PointTarget>& tgt,
                                   Matrix4& T) const override {
    if (src.size() != tgt.size()) return;  // "Number or points in source differs than target!": T untouched
    run(src.points.data(), tgt.points.data(), src.size(), T);
  }
  void estimateRigidTransformation(const pcl::PointCloud<PointSource>& src, const pcl::Indices& is,
                                   const pcl::PointCloud<PointTarget>& tgt, Matrix4& T) const override {
    if (is.size() != tgt.size()) return;
    std::vector<PointSource> s;
    for (pcl::index_t i : is) s.push_back(src[std::size_t(i)]);
    run(s.data(), tgt.points.data(), s.size(), T);
  }
  void estimateRigidTransformation(const pcl::PointCloud<PointSource>& src, const pcl::Indices& is,
                                   const pcl::PointCloud<PointTarget>& tgt, const pcl::Indices& it, Matrix4& T) const override {
    if (is.size() != it.size()) return;
    std::vector<PointSource> s;
    std::vector<PointTarget> t;
    for (pcl::index_t i : is) s.push_back(src[std::size_t(i)]);
    for (pcl::index_t i : it) t.push_back(tgt[std::size_t(i)]);
    run(s.data(), t.data(), s.size(), T);
  }
  void estimateRigidTransformation(const pcl::PointCloud<PointSource>& src, const pcl::PointCloud<PointTarget>& tgt,
                                   const pcl::Correspondences& corr, Matrix4& T) const override {
    std::vector<PointSource> s;
    std::vector<PointTarget> t;
    for (const pcl::Correspondence& c : corr) {
      s.push_back(src[std::size_t(c.index_query)]);
      t.push_back(tgt[std::size_t(c.index_match)]);
    }
    run(s.data(), t.data(), s.size(), T);
  }
 private:
  void run(const PointSource* s, const PointTarget* t, std::size_t n, Matrix4& T) const {
    if (!dev_ || !dev_->ok() || n == 0) return;
    static_assert(MODE == PCLHIP_ICP_POINT_TO_POINT || (record_fits_normals<PointTarget>() &&
                  (MODE != PCLHIP_ICP_SYMMETRIC || record_fits_normals<PointSource>())),
                  "point-to-plane estimators need normals in the target (and, symmetric, in the source) point type");
    if (!record_layout<PointSource>().xyz_at_0 || !record_layout<PointTarget>().xyz_at_0) return;
    if (MODE != PCLHIP_ICP_POINT_TO_POINT && !has_normal_fields<PointTarget>()) return;  // normals elsewhere in the record
    if (MODE == PCLHIP_ICP_SYMMETRIC && !has_normal_fields<PointSource>()) return;
    const char* sb = reinterpret_cast<const char*>(s);
    const char* tb = reinterpret_cast<const char*>(t);
    float m[16];
    if (pclhip_estimate_rigid_transformation(dev_->get(), MODE, sb, sizeof(PointSource),
                                             has_normal_fields<PointSource>() ? sb + 16 : nullptr, sizeof(PointSource), tb,
                                             sizeof(PointTarget), has_normal_fields<PointTarget>() ? tb + 16 : nullptr,
                                             sizeof(PointTarget), n, enforce_ ? 1 : 0, m, nullptr) != PCLHIP_OK)
      return;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) T(r, c) = Scalar(m[4 * r + c]);
  }
  Device::Ptr dev_;
  bool enforce_ = true;
};
template <typename S, typename T, typename Scalar = float>
using TransformationEstimationSVDHIP = TransformationEstimationHIP<S, T, PCLHIP_ICP_POINT_TO_POINT, Scalar>;
template <typename S, typename T, typename Scalar = float>
using TransformationEstimationPointToPlaneLLSHIP = TransformationEstimationHIP<S, T, PCLHIP_ICP_POINT_TO_PLANE, Scalar>;
template <typename S, typename T, typename Scalar = float>
using TransformationEstimationSymmetricPointToPlaneLLSHIP = TransformationEstimationHIP<S, T, PCLHIP_ICP_SYMMETRIC, Scalar>;

}  // namespace plugin
}  // namespace pclhip
