// The real-PCL binding (include/pclhip/pcl_plugin.hpp) compiled against the PCL mock (tests/cpp/pcl_mock: the
// PCL base classes with their real signatures, no Eigen) and driven THROUGH THE BASE-CLASS INTERFACES, the way
// a PCL application and PCL's own algorithms reach a plugin:
//   pcl::search::Search<PointT>*                 -> KdTreeHIP            (search.h:144-273 virtuals)
//   pcl::registration::CorrespondenceEstimationBase*  -> CorrespondenceEstimationHIP (correspondence_estimation.h:277-293)
//   pcl::Registration<...>* / pcl::IterativeClosestPoint* -> IterativeClosestPoint[WithNormals]HIP
//        (registration.h:678-679 computeTransformation, reached from PCL's own align())
// Goldens: test/registration/test_registration_api.cpp:83-104 (397 bunny correspondences),
// test/registration/test_registration.cpp:236-270 (ICP 4x4 @1e-3), test/kdtree/test_kdtree.cpp:226-289.
// Inputs are written by tests/test_gpu_cpp_adapters.py from tests/golden/.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <memory>
#include <thread>

#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

#include "pclhip/pcl_plugin.hpp"

using namespace pclhip::plugin;

static int failures = 0;
#define EXPECT(cond)                                                        \
  do {                                                                      \
    if (!(cond)) {                                                          \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      ++failures;                                                           \
    }                                                                       \
  } while (0)

template <typename PointT>
static typename pcl::PointCloud<PointT>::Ptr load_xyz(const char* path) {
  typename pcl::PointCloud<PointT>::Ptr c(new pcl::PointCloud<PointT>);
  std::ifstream f(path);
  float x, y, z;
  while (f >> x >> y >> z) {
    PointT p;
    p.x = x; p.y = y; p.z = z;
    c->push_back(p);
  }
  return c;
}

// a TransformationEstimation the binding has never heard of: the subclass must hand over to PCL's own loop
template <typename S, typename T>
struct ForeignEstimation : pcl::registration::TransformationEstimation<S, T, float> {
  using Matrix4 = typename pcl::registration::TransformationEstimation<S, T, float>::Matrix4;
  void estimateRigidTransformation(const pcl::PointCloud<S>&, const pcl::PointCloud<T>&, Matrix4&) const override {}
  void estimateRigidTransformation(const pcl::PointCloud<S>&, const pcl::Indices&, const pcl::PointCloud<T>&, Matrix4&) const override {}
  void estimateRigidTransformation(const pcl::PointCloud<S>&, const pcl::Indices&, const pcl::PointCloud<T>&,
                                   const pcl::Indices&, Matrix4&) const override {}
  void estimateRigidTransformation(const pcl::PointCloud<S>&, const pcl::PointCloud<T>&, const pcl::Correspondences&,
                                   Matrix4&) const override {}
};

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  auto dev = std::make_shared<Device>(0);
  if (!dev->ok()) {
    std::fprintf(stderr, "no device: %s\n", pclhip_last_error(nullptr));
    return 3;
  }
  auto source = load_xyz<pcl::PointXYZ>(argv[1]);
  auto target = load_xyz<pcl::PointXYZ>(argv[2]);
  EXPECT(source->size() == 397 && target->size() == 361);
  std::vector<int> gold;
  {
    std::ifstream f(argv[3]);
    int a, b;
    while (f >> a >> b) gold.push_back(b);
  }
  EXPECT(gold.size() == 397);

  {  // 1. the search backend through pcl::search::Search<PointT>
    pcl::search::Search<pcl::PointXYZ>::Ptr search(new KdTreeHIP<pcl::PointXYZ>(dev));
    EXPECT(search->getName() == "KdTreeHIP");
    EXPECT(search->setInputCloud(target));
    EXPECT(search->getInputCloud() == target);
    pcl::Indices idx;
    std::vector<float> d2;
    EXPECT(search->nearestKSearch((*source)[0], 5, idx, d2) == 5);
    EXPECT(idx.size() == 5 && idx[0] == gold[0] && d2[0] <= d2[1] && d2[1] <= d2[4]);
    EXPECT(search->nearestKSearch(*source, 0, 5, idx, d2) == 5 && idx[0] == gold[0]);  // (cloud, index) overload of the base
    EXPECT(search->nearestKSearch((*source)[0], 1000, idx, d2) == 361);                 // k clamped (kdtree_flann.hpp:241-242)
    std::vector<pcl::Indices> bi;
    std::vector<std::vector<float>> bd;
    search->nearestKSearch(*source, pcl::Indices(), 1, bi, bd);                         // the batch virtual: one launch
    EXPECT(bi.size() == 397);
    for (std::size_t i = 0; i < bi.size() && i < gold.size(); ++i) EXPECT(bi[i].size() == 1 && bi[i][0] == gold[i]);
    pcl::Indices ridx;
    std::vector<float> rd2;
    const int nr = search->radiusSearch((*source)[0], 0.02, ridx, rd2);
    EXPECT(nr > 0 && int(ridx.size()) == nr && ridx[0] == gold[0]);
    for (int i = 1; i < nr; ++i) EXPECT(rd2[std::size_t(i) - 1] <= rd2[std::size_t(i)] && rd2[std::size_t(i)] < 0.02f * 0.02f);
    EXPECT(search->radiusSearch((*source)[0], 0.02, ridx, rd2, 3) == (nr < 3 ? nr : 3));
    // indices: results refer to the original cloud
    pcl::IndicesPtr sub(new pcl::Indices);
    for (int i = 0; i < 361; i += 2) sub->push_back(i);
    EXPECT(search->setInputCloud(target, sub));
    EXPECT(search->nearestKSearch((*source)[0], 3, idx, d2) == 3);
    for (int v : idx) EXPECT(v % 2 == 0);
  }

  {  // 2. point representations (test/kdtree/test_kdtree.cpp:252-282 uses these two)
    auto tree = std::make_shared<KdTreeHIP<pcl::PointXYZ>>(dev);
    pcl::search::KdTree<pcl::PointXYZ>& base = *tree;
    std::shared_ptr<pcl::CustomPointRepresentation<pcl::PointXYZ>> xy(new pcl::CustomPointRepresentation<pcl::PointXYZ>(2, 0));
    base.setPointRepresentation(xy);
    EXPECT(tree->representationSupported());
    EXPECT(base.setInputCloud(target));
    pcl::Indices idx;
    std::vector<float> d2;
    pcl::PointXYZ q = (*target)[17];
    q.z += 5.0f;  // z is not part of the representation: the point still finds itself at distance 0
    EXPECT(base.nearestKSearch(q, 1, idx, d2) == 1 && idx[0] == 17 && d2[0] == 0.0f);
    std::shared_ptr<pcl::DefaultPointRepresentation<pcl::PointXYZ>> scaled(new pcl::DefaultPointRepresentation<pcl::PointXYZ>);
    const float alpha[3] = {1.0f, 2.0f, 3.0f};
    scaled->setRescaleValues(alpha);
    base.setPointRepresentation(scaled);
    pcl::PointXYZ r = (*target)[17];
    r.y += 0.001f;
    EXPECT(base.nearestKSearch(r, 1, idx, d2) == 1 && idx[0] == 17);
    EXPECT(std::fabs(d2[0] - 4.0f * 0.001f * 0.001f) < 1e-8f);  // distances live in the rescaled space
  }

  {  // 3. the batch correspondence entry through CorrespondenceEstimationBase*
    pcl::registration::CorrespondenceEstimationBase<pcl::PointXYZ, pcl::PointXYZ, float>::Ptr ce(
        new CorrespondenceEstimationHIP<pcl::PointXYZ, pcl::PointXYZ, float>);
    ce->setSearchMethodTarget(std::make_shared<KdTreeHIP<pcl::PointXYZ>>(dev));
    ce->setInputSource(source);
    ce->setInputTarget(target);
    pcl::Correspondences corr;
    ce->determineCorrespondences(corr);
    EXPECT(corr.size() == 397);
    for (std::size_t i = 0; i < corr.size() && i < gold.size(); ++i)
      EXPECT(corr[i].index_query == int(i) && corr[i].index_match == gold[i]);
    pcl::Correspondences rec;
    ce->determineReciprocalCorrespondences(rec);
    EXPECT(rec.size() == 53);  // test_registration_api_data.h:404-459
    auto copy = ce->clone();   // clone(): Registration copies the estimator for worker threads
    pcl::Correspondences corr2;
    copy->determineCorrespondences(corr2, 0.01);
    EXPECT(!corr2.empty() && corr2.size() < 397);
    for (const auto& c : corr2) EXPECT(c.distance <= 0.01f * 0.01f && c.index_match == gold[std::size_t(c.index_query)]);
    // PCL's own per-point estimator over the same backend (one launch per point): identical pairs
    pcl::registration::CorrespondenceEstimation<pcl::PointXYZ, pcl::PointXYZ, float> stock;
    stock.setSearchMethodTarget(std::make_shared<KdTreeHIP<pcl::PointXYZ>>(dev));
    stock.setInputSource(source);
    stock.setInputTarget(target);
    pcl::Correspondences slow;
    stock.determineCorrespondences(slow);
    EXPECT(slow.size() == corr.size());
    for (std::size_t i = 0; i < slow.size() && i < corr.size(); ++i)
      EXPECT(slow[i].index_match == corr[i].index_match && slow[i].distance == corr[i].distance);
    // setIndicesSource: only those source points
    pcl::IndicesPtr some(new pcl::Indices{3, 10, 200, 396});
    ce->setIndicesSource(some);
    ce->determineCorrespondences(corr);
    EXPECT(corr.size() == 4 && corr[0].index_query == 3 && corr[3].index_query == 396 && corr[2].index_match == gold[200]);
    // one correspondence per ENTRY of the list, in the list's order, duplicates included
    // (impl/correspondence_estimation.hpp:160-216 loops over indices_)
    pcl::IndicesPtr shuffled(new pcl::Indices{396, 3, 200, 3, 10});
    ce->setIndicesSource(shuffled);
    ce->determineCorrespondences(corr);
    bool order = corr.size() == 5;
    for (std::size_t i = 0; order && i < 5; ++i)
      order = corr[i].index_query == (*shuffled)[i] && corr[i].index_match == gold[std::size_t((*shuffled)[i])];
    EXPECT(order);
    // a list as long as the cloud is still a list: a permutation comes back in its own order
    pcl::IndicesPtr reversed(new pcl::Indices);
    for (int i = 396; i >= 0; --i) reversed->push_back(i);
    ce->setIndicesSource(reversed);
    ce->determineCorrespondences(corr);
    EXPECT(corr.size() == 397 && corr.front().index_query == 396 && corr.back().index_query == 0 &&
           corr.front().index_match == gold[396]);
  }

  {  // 3b. the `const` search virtuals under CONCURRENT callers, as PCL itself calls them: the stock per-point estimator
     //     with setNumberOfThreads(8) (impl/correspondence_estimation.hpp:163-175), NormalEstimationOMP's loop
     //     (features/.../impl/normal_3d_omp.hpp:76-81) and the default batch radiusSearch (search/.../impl/search.hpp:164-190)
     //     all run nearestKSearch / radiusSearch of ONE tree from several threads
    auto tree = std::make_shared<KdTreeHIP<pcl::PointXYZ>>(dev);
    pcl::registration::CorrespondenceEstimation<pcl::PointXYZ, pcl::PointXYZ, float> stock;
    stock.setNumberOfThreads(8);
    stock.setSearchMethodTarget(tree);
    stock.setInputSource(source);
    stock.setInputTarget(target);
    for (int round = 0; round < 3; ++round) {
      pcl::Correspondences par;
      stock.determineCorrespondences(par);
      EXPECT(par.size() == 397);
      bool same = par.size() == gold.size();
      for (std::size_t i = 0; same && i < par.size(); ++i) same = par[i].index_query == int(i) && par[i].index_match == gold[i];
      EXPECT(same);  // the 397 golden pairs (test_registration_api_data.h:3-402), whatever the interleaving
    }
    // plain threads (no OpenMP needed): k-NN and radius searches of the same tree mixed, every answer equal to the
    // serial one
    const pcl::search::Search<pcl::PointXYZ>& search = *tree;
    std::vector<pcl::Indices> want_k(source->size()), want_r(source->size());
    std::vector<std::vector<float>> want_kd(source->size()), want_rd(source->size());
    for (std::size_t i = 0; i < source->size(); ++i) {
      search.nearestKSearch((*source)[i], 4, want_k[i], want_kd[i]);
      search.radiusSearch((*source)[i], 0.01, want_r[i], want_rd[i]);
    }
    std::atomic<int> bad{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < 8; ++t)
      pool.emplace_back([&, t] {
        pcl::Indices idx;
        std::vector<float> d2;
        for (std::size_t i = std::size_t(t); i < source->size(); i += 8) {
          if (search.nearestKSearch((*source)[i], 4, idx, d2) != 4 || idx != want_k[i] || d2 != want_kd[i]) ++bad;
          if (search.radiusSearch((*source)[i], 0.01, idx, d2) != int(want_r[i].size()) || idx != want_r[i] || d2 != want_rd[i]) ++bad;
        }
      });
    for (auto& th : pool) th.join();
    EXPECT(bad.load() == 0);
    // the batch radiusSearch virtual (search.h:349-355) is ONE call here; same lists as the per-point virtual
    std::vector<pcl::Indices> br;
    std::vector<std::vector<float>> brd;
    search.radiusSearch(*source, pcl::Indices(), 0.01, br, brd);
    EXPECT(br.size() == source->size());
    bool lists = br.size() == source->size();
    for (std::size_t i = 0; lists && i < br.size(); ++i) lists = br[i] == want_r[i] && brd[i] == want_rd[i];
    EXPECT(lists);
    pcl::Indices pick{5, 17, 396, 17};
    search.radiusSearch(*source, pick, 0.01, br, brd, 3);  // an index list, max_nn = 3
    EXPECT(br.size() == 4);
    for (std::size_t j = 0; j < br.size() && j < pick.size(); ++j) {
      const auto& full = want_r[std::size_t(pick[j])];
      const std::size_t m = full.size() < 3 ? full.size() : 3;
      EXPECT(br[j].size() == m && std::equal(br[j].begin(), br[j].end(), full.begin()));
    }
  }

  // test/registration/test_registration.cpp:251-269 (tests/golden/golden.json: icp_bunny)
  const double G[16] = {0.8806, 0.036481287, -0.4724, 0.03453, -0.02354, 0.9992, 0.03326, -0.001519,
                        0.4732, -0.01817, 0.8808, 0.04116, 0, 0, 0, 1};
  float T_point[16];
  {  // 4. IterativeClosestPoint through pcl::Registration*: PCL's align() calls the overridden computeTransformation
    pcl::Registration<pcl::PointXYZ, pcl::PointXYZ, float>::Ptr reg(new IterativeClosestPointHIP<pcl::PointXYZ, pcl::PointXYZ>(dev));
    reg->setInputSource(source);
    reg->setInputTarget(target);
    reg->setMaximumIterations(50);
    reg->setTransformationEpsilon(1e-8);
    reg->setMaxCorrespondenceDistance(0.05);
    pcl::PointCloud<pcl::PointXYZ> out;
    reg->align(out);
    auto* hip = dynamic_cast<IterativeClosestPointHIP<pcl::PointXYZ, pcl::PointXYZ>*>(reg.get());
    EXPECT(hip != nullptr && hip->deferredReason().empty());
    EXPECT(reg->hasConverged() && out.size() == source->size());
    const auto T = reg->getFinalTransformation();
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) {
        EXPECT(std::fabs(T(r, c) - G[4 * r + c]) < ((r == 0 && c == 1) ? 1e-2 : 1e-3));
        T_point[4 * r + c] = T(r, c);
      }
    // the output is the input moved by the final transformation
    const auto& p = (*source)[5];
    const float x = T(0, 0) * p.x + T(0, 1) * p.y + T(0, 2) * p.z + T(0, 3);
    EXPECT(std::fabs(out[5].x - x) < 1e-5f);
    EXPECT(reg->getFitnessScore() < 0.001);  // test_registration.cpp:301-302
    // registration.h:450-452 is not virtual: through pcl::Registration* that was PCL's own loop over the HIP search
    // backend (one nearestKSearch per point); on the HIP class it is one launch -- the same score
    const double via_base = reg->getFitnessScore(), on_device = hip->getFitnessScore();
    EXPECT(std::fabs(via_base - on_device) <= 1e-4 * via_base);
    EXPECT(std::fabs(reg->getFitnessScore(1e-5) - hip->getFitnessScore(1e-5)) <= 1e-4 * via_base);  // max_range as given
    auto* icp = dynamic_cast<pcl::IterativeClosestPoint<pcl::PointXYZ, pcl::PointXYZ, float>*>(reg.get());
    EXPECT(icp->getConvergeCriteria()->getConvergenceState() !=
           pcl::registration::DefaultConvergenceCriteria<float>::CONVERGENCE_CRITERIA_NOT_CONVERGED);
    // rejectors and reciprocal correspondences map onto the device chain
    std::shared_ptr<pcl::registration::CorrespondenceRejectorMedianDistance> med(
        new pcl::registration::CorrespondenceRejectorMedianDistance);
    med->setMedianFactor(4.0);
    reg->addCorrespondenceRejector(med);
    icp->setUseReciprocalCorrespondences(true);
    reg->align(out);
    EXPECT(hip->deferredReason().empty() && reg->hasConverged());
    EXPECT(reg->getFitnessScore() < 0.01);  // a different (filtered) objective: a registration, not the golden pose
    // a source subset through PCLBase::setIndices
    reg->clearCorrespondenceRejectors();
    icp->setUseReciprocalCorrespondences(false);
    pcl::IndicesPtr half(new pcl::Indices);
    for (int i = 0; i < 397; i += 2) half->push_back(i);
    reg->setIndices(half);
    reg->align(out);
    EXPECT(hip->deferredReason().empty() && reg->hasConverged() && out.size() == source->size());
    const auto T3 = reg->getFinalTransformation();
    for (int r = 0; r < 3; ++r) EXPECT(std::fabs(T3(r, 3) - G[4 * r + 3]) < 1e-2);  // half the points: the same pose, roughly
    EXPECT(reg->getFitnessScore() < 0.001);
    // use_indices (impl/registration.hpp:141-144): the device holds the subset the alignment ran on -> scores it there;
    // the whole cloud is then PCL's loop; both equal what the base computes
    EXPECT(std::fabs(hip->getFitnessScore(1e30, true) - reg->getFitnessScore(1e30, true)) <= 1e-4 * reg->getFitnessScore(1e30, true));
    EXPECT(hip->getFitnessScore(1e30, false) == reg->getFitnessScore(1e30, false));
    // another list of the SAME size (PCLBase::setIndices does not flag the source as updated): the device must not keep
    // the old subset -- twenty points near one end of the scan cannot give the pose of the whole scan
    pcl::IndicesPtr other(new pcl::Indices);
    for (int i = 0; i < 199; ++i) other->push_back(i < 20 ? i : 19);
    reg->setIndices(other);
    reg->align(out);
    const auto T4 = reg->getFinalTransformation();
    bool differs = false;
    for (int r = 0; r < 3; ++r) differs = differs || T4(r, 3) != T3(r, 3);
    EXPECT(hip->deferredReason().empty() && differs);
    // the target cloud modified in place and set again: the index is rebuilt (often at the freed address) and the
    // registration must follow the rebuild, not the address
    reg->setIndices(half);
    reg->setInputTarget(target);
    reg->align(out);
    const auto T5 = reg->getFinalTransformation();
    for (int r = 0; r < 3; ++r) EXPECT(T5(r, 3) == T3(r, 3));
  }

  {  // 5. what the binding cannot express is handed to PCL's own loop -- never silently replaced.  (The mock has
     //    no CPU loop: it aborts there, so only the decision is checked, through a subclass that records it.)
    struct Probe : IterativeClosestPointHIP<pcl::PointXYZ, pcl::PointXYZ> {
      using IterativeClosestPointHIP<pcl::PointXYZ, pcl::PointXYZ>::IterativeClosestPointHIP;
      bool foreign() const { return this->estimatorKind() == this->FOREIGN; }
    } probe(dev);
    EXPECT(!probe.foreign());
    probe.setTransformationEstimation(std::make_shared<ForeignEstimation<pcl::PointXYZ, pcl::PointXYZ>>());
    EXPECT(probe.foreign());
  }

  {  // 6. IterativeClosestPointWithNormals on PointNormal clouds (normals read from the records, +16 bytes)
    auto tn = load_xyz<pcl::PointNormal>(argv[2]);
    auto sn = load_xyz<pcl::PointNormal>(argv[1]);
    std::ifstream fn(argv[4]);  // target normals (nx ny nz per line), computed by the Python side
    for (auto& p : tn->points) fn >> p.normal_x >> p.normal_y >> p.normal_z;
    std::shared_ptr<pcl::IterativeClosestPointWithNormals<pcl::PointNormal, pcl::PointNormal, float>> reg(
        new IterativeClosestPointWithNormalsHIP<pcl::PointNormal, pcl::PointNormal>(dev));
    reg->setInputSource(sn);
    reg->setInputTarget(tn);
    reg->setMaximumIterations(50);
    reg->setTransformationEpsilon(1e-8);
    reg->setMaxCorrespondenceDistance(0.05);
    pcl::PointCloud<pcl::PointNormal> out;
    reg->align(out);
    auto* hip = dynamic_cast<IterativeClosestPointWithNormalsHIP<pcl::PointNormal, pcl::PointNormal>*>(reg.get());
    EXPECT(hip != nullptr && hip->deferredReason().empty());
    EXPECT(reg->hasConverged());
    const auto T = reg->getFinalTransformation();
    // the fitness bar of test_registration.cpp:272-318, and roughly the pose point-to-point found
    EXPECT(reg->getFitnessScore() < 0.001);
    for (int r = 0; r < 3; ++r) {
      EXPECT(std::fabs(T(r, 3) - T_point[4 * r + 3]) < 2e-2);
      for (int c = 0; c < 3; ++c) EXPECT(std::fabs(T(r, c) - T_point[4 * r + c]) < 1e-1);
    }
    EXPECT(!reg->getUseSymmetricObjective());
    // a search tree built BEFORE the cloud had its normals, handed over with force_no_recompute -- the
    // alignment must use the normals the cloud has NOW (as PCL's CPU path would read them), not the ones the tree's own
    // upload happened to see
    pcl::PointCloud<pcl::PointNormal>::Ptr late(new pcl::PointCloud<pcl::PointNormal>(*tn));
    for (auto& p : late->points) { p.normal_x = 1.0f; p.normal_y = 0.0f; p.normal_z = 0.0f; }   // placeholders
    auto early_tree = std::make_shared<KdTreeHIP<pcl::PointNormal>>(dev);
    EXPECT(early_tree->setInputCloud(late));
    for (std::size_t i = 0; i < late->size(); ++i) {                                             // the real ones arrive
      (*late)[i].normal_x = (*tn)[i].normal_x; (*late)[i].normal_y = (*tn)[i].normal_y; (*late)[i].normal_z = (*tn)[i].normal_z;
    }
    IterativeClosestPointWithNormalsHIP<pcl::PointNormal, pcl::PointNormal> reg2(dev);
    reg2.setInputSource(sn);
    reg2.setInputTarget(late);
    reg2.setSearchMethodTarget(early_tree, true);
    reg2.setMaximumIterations(50);
    reg2.setTransformationEpsilon(1e-8);
    reg2.setMaxCorrespondenceDistance(0.05);
    pcl::PointCloud<pcl::PointNormal> out2;
    reg2.align(out2);
    EXPECT(reg2.deferredReason().empty() && reg2.hasConverged());
    const auto T2 = reg2.getFinalTransformation();
    bool same_pose = true;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) same_pose = same_pose && T2(r, c) == T(r, c);
    EXPECT(same_pose);
  }

  {  // 6b. Scalar = double (registration.h:56, icp.h:97,339): the same registration with the poses kept in double
    auto tn = load_xyz<pcl::PointNormal>(argv[2]);
    auto sn = load_xyz<pcl::PointNormal>(argv[1]);
    std::ifstream fn(argv[4]);
    for (auto& p : tn->points) fn >> p.normal_x >> p.normal_y >> p.normal_z;
    IterativeClosestPointWithNormalsHIP<pcl::PointNormal, pcl::PointNormal, float> regf(dev);
    IterativeClosestPointWithNormalsHIP<pcl::PointNormal, pcl::PointNormal, double> regd(dev);
    pcl::Registration<pcl::PointNormal, pcl::PointNormal, double>& base_d = regd;  // what a double pipeline holds
    pcl::PointCloud<pcl::PointNormal> outf, outd;
    regf.setInputSource(sn); regf.setInputTarget(tn); regf.setMaximumIterations(50);
    regf.setTransformationEpsilon(1e-8); regf.setMaxCorrespondenceDistance(0.05);
    base_d.setInputSource(sn); base_d.setInputTarget(tn); base_d.setMaximumIterations(50);
    base_d.setTransformationEpsilon(1e-8); base_d.setMaxCorrespondenceDistance(0.05);
    regf.align(outf);
    Eigen::Matrix<double, 4, 4> guess = Eigen::Matrix<double, 4, 4>::Identity();
    base_d.align(outd, guess);
    EXPECT(regd.deferredReason().empty() && base_d.hasConverged() && regd.iterations() == regf.iterations());
    const Eigen::Matrix<double, 4, 4> Td = base_d.getFinalTransformation();
    const Eigen::Matrix4f Tf = regf.getFinalTransformation();
    bool same = true;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) same = same && Td(r, c) == double(Tf(r, c));
    EXPECT(same);
    EXPECT(outd.size() == outf.size() && outd[7].x == outf[7].x && outd[7].normal_z == outf[7].normal_z);
    // the estimators and the point-to-point registration instantiate for double too
    TransformationEstimationPointToPlaneLLSHIP<pcl::PointNormal, pcl::PointNormal, double> ted(dev);
    TransformationEstimationPointToPlaneLLSHIP<pcl::PointNormal, pcl::PointNormal, float> tef(dev);
    Eigen::Matrix<double, 4, 4> Ed = Eigen::Matrix<double, 4, 4>::Identity();
    Eigen::Matrix4f Ef = Eigen::Matrix4f::Identity();
    pcl::PointCloud<pcl::PointNormal> head_s, head_t;
    for (std::size_t i = 0; i < 300; ++i) { head_s.push_back((*tn)[i]); head_t.push_back((*tn)[i]); head_s[i].x += 0.001f; }
    ted.estimateRigidTransformation(head_s, head_t, Ed);
    tef.estimateRigidTransformation(head_s, head_t, Ef);
    EXPECT(Ed(0, 3) == double(Ef(0, 3)) && std::fabs(Ed(0, 3) + 0.001) < 1e-4);
    IterativeClosestPointHIP<pcl::PointXYZ, pcl::PointXYZ, double> regp(dev);
    auto sx = load_xyz<pcl::PointXYZ>(argv[1]);
    auto tx = load_xyz<pcl::PointXYZ>(argv[2]);
    regp.setInputSource(sx); regp.setInputTarget(tx); regp.setMaximumIterations(50); regp.setMaxCorrespondenceDistance(0.05);
    pcl::PointCloud<pcl::PointXYZ> outp;
    regp.align(outp);
    EXPECT(regp.deferredReason().empty() && regp.hasConverged());
    for (int r = 0; r < 3; ++r) EXPECT(std::fabs(regp.getFinalTransformation()(r, 3) - double(T_point[4 * r + 3])) < 1e-3);
  }

  {  // 7. NormalEstimationHIP through pcl::Feature::compute (feature.hpp:195-229 -> the overridden computeFeature)
    auto tgt = load_xyz<pcl::PointXYZ>(argv[2]);
    auto src = load_xyz<pcl::PointXYZ>(argv[1]);
    std::shared_ptr<pcl::NormalEstimation<pcl::PointXYZ, pcl::Normal>> ne(new NormalEstimationHIP<pcl::PointXYZ, pcl::Normal>(dev));
    pcl::Feature<pcl::PointXYZ, pcl::Normal>& feature = *ne;  // what a PCL pipeline holds
    ne->setViewPoint(0.0f, 0.0f, 10.0f);
    feature.setInputCloud(tgt);
    feature.setKSearch(10);
    pcl::PointCloud<pcl::Normal> self, sub, cross;
    feature.compute(self);
    EXPECT(self.size() == tgt->size() && self.is_dense);
    std::ifstream fn(argv[4]);  // the Python side's k = 10 normals of the same cloud, same view point
    bool same = true;
    for (std::size_t i = 0; i < self.size(); ++i) {
      float nx, ny, nz;
      fn >> nx >> ny >> nz;
      same = same && std::fabs(self[i].normal_x * nx + self[i].normal_y * ny + self[i].normal_z * nz) > 1.0f - 1e-5f;
    }
    EXPECT(same);
    pcl::IndicesPtr every_third(new pcl::Indices);
    for (std::size_t i = 0; i < tgt->size(); i += 3) every_third->push_back(pcl::index_t(i));
    feature.setIndices(every_third);   // PCLBase::setIndices: one normal per index
    feature.compute(sub);
    EXPECT(sub.size() == every_third->size() && sub.width == sub.size() && sub.height == 1);
    bool equal = true;
    for (std::size_t j = 0; j < sub.size(); ++j) {
      const pcl::Normal& a = sub[j];
      const pcl::Normal& b = self[std::size_t((*every_third)[j])];
      equal = equal && a.normal_x == b.normal_x && a.normal_y == b.normal_y && a.normal_z == b.normal_z && a.curvature == b.curvature;
    }
    EXPECT(equal);
    std::shared_ptr<pcl::NormalEstimation<pcl::PointXYZ, pcl::Normal>> ne2(new NormalEstimationHIP<pcl::PointXYZ, pcl::Normal>(dev));
    ne2->setViewPoint(0.0f, 0.0f, 10.0f);
    ne2->setInputCloud(src);          // normals AT the source points ...
    ne2->setSearchSurface(tgt);       // ... from the target surface (Feature::setSearchSurface)
    ne2->setRadiusSearch(0.03);
    ne2->compute(cross);
    EXPECT(cross.size() == src->size());
    int finite = 0;
    bool unit = true;
    for (std::size_t j = 0; j < cross.size(); ++j) {
      if (std::isnan(cross[j].normal_x)) continue;
      ++finite;
      const float l = cross[j].normal_x * cross[j].normal_x + cross[j].normal_y * cross[j].normal_y + cross[j].normal_z * cross[j].normal_z;
      unit = unit && std::fabs(l - 1.0f) < 1e-4f && cross[j].normal_z * (10.0f - (*src)[j].z) + cross[j].normal_x * (0.0f - (*src)[j].x) +
                                                          cross[j].normal_y * (0.0f - (*src)[j].y) >= 0.0f;
    }
    EXPECT(finite > int(cross.size()) / 2 && unit);
    auto* hip = dynamic_cast<NormalEstimationHIP<pcl::PointXYZ, pcl::Normal>*>(ne2.get());
    EXPECT(hip != nullptr && hip->deferredReason().empty());
    ne2->setKSearch(5);               // both K and radius: Feature::initCompute refuses (feature.hpp:131-140)
    ne2->compute(cross);
    EXPECT(cross.size() == 0);
  }

  {  // 8. VoxelGridHIP through pcl::Filter::filter (filter.h:121-144 -> the overridden applyFilter)
    auto src = load_xyz<pcl::PointXYZ>(argv[1]);
    std::shared_ptr<pcl::VoxelGrid<pcl::PointXYZ>> grid(new VoxelGridHIP<pcl::PointXYZ>(dev));
    pcl::Filter<pcl::PointXYZ>& filter = *grid;
    grid->setLeafSize(0.02f, 0.02f, 0.02f);
    grid->setSaveLeafLayout(true);
    filter.setInputCloud(src);
    pcl::PointCloud<pcl::PointXYZ> out;
    filter.filter(out);
    EXPECT(out.size() == 103 && out.width == 103 && out.height == 1 && out.is_dense);  // test/filters/test_filters.cpp:566-596
    const auto div = grid->getNrDivisions();
    EXPECT(grid->getLeafLayout().size() == std::size_t(div[0]) * std::size_t(div[1]) * std::size_t(div[2]));
    const int c0 = grid->getCentroidIndex((*src)[0]);
    EXPECT(c0 >= 0 && c0 < 103);
    // the pass-through filter and its negative, PCL's own setters: the two halves come from disjoint sets of points
    grid->setSaveLeafLayout(false);
    grid->setFilterFieldName("y");
    grid->setFilterLimits(0.10, 1.0);
    pcl::PointCloud<pcl::PointXYZ> upper, lower;
    filter.filter(upper);
    grid->setFilterLimitsNegative(true);
    filter.filter(lower);
    bool split = upper.size() > 0 && lower.size() > 0;
    for (const auto& p : upper.points) split = split && p.y >= 0.10f;
    for (const auto& p : lower.points) split = split && p.y <= 0.10f;
    EXPECT(split);
    grid->setFilterFieldName("curvature");  // not a field of pcl::PointXYZ
    filter.filter(out);
    EXPECT(out.size() == 0);
    // cloud_in == cloud_out (filter.h:126-134)
    pcl::PointCloud<pcl::PointXYZ>::Ptr inplace(new pcl::PointCloud<pcl::PointXYZ>(*src));
    grid->setFilterFieldName("");
    filter.setInputCloud(inplace);
    filter.filter(*inplace);
    EXPECT(inplace->size() == 103);
    // pcl::PointNormal: all fields, then coordinates only (voxel_grid.h:293-301)
    auto srcn = load_xyz<pcl::PointNormal>(argv[1]);
    for (auto& p : srcn->points) { p.normal_z = 1.0f; p.curvature = 0.5f; }
    VoxelGridHIP<pcl::PointNormal> gridn(dev);
    gridn.setLeafSize(0.02f, 0.02f, 0.02f);
    gridn.setInputCloud(srcn);
    pcl::PointCloud<pcl::PointNormal> all, xyz_only;
    gridn.filter(all);
    gridn.setDownsampleAllData(false);
    gridn.filter(xyz_only);
    bool fields = all.size() == 103 && xyz_only.size() == 103;
    for (std::size_t i = 0; fields && i < all.size(); ++i)
      fields = all[i].normal_z == 1.0f && all[i].curvature == 0.5f && xyz_only[i].normal_z == 0.0f && xyz_only[i].curvature == 0.0f &&
               all[i].x == xyz_only[i].x;
    EXPECT(fields);
    // point types the device path must hand back to pcl::VoxelGrid when every field is to be averaged: PointXYZI (a field
    // the device would drop), PointXYZRGBNormal (48 bytes like PointNormal, but rgb at +32 and the curvature at +36)
    using pclhip::plugin::record_layout;
    EXPECT(record_layout<pcl::PointXYZ>().xyz_at_0 && !record_layout<pcl::PointXYZ>().other_fields &&
           !record_layout<pcl::PointXYZ>().normal_at_16);
    EXPECT(record_layout<pcl::PointNormal>().normal_at_16 && record_layout<pcl::PointNormal>().curvature_at_32 &&
           !record_layout<pcl::PointNormal>().other_fields);
    EXPECT(record_layout<pcl::PointXYZI>().xyz_at_0 && record_layout<pcl::PointXYZI>().other_fields &&
           !pclhip::plugin::has_normal_fields<pcl::PointXYZI>());
    EXPECT(record_layout<pcl::PointXYZRGBNormal>().normal_at_16 && !record_layout<pcl::PointXYZRGBNormal>().curvature_at_32 &&
           record_layout<pcl::PointXYZRGBNormal>().other_fields);
    auto srci = load_xyz<pcl::PointXYZI>(argv[1]);
    for (auto& p : srci->points) p.intensity = 7.0f;
    VoxelGridHIP<pcl::PointXYZI> gridi(dev);
    gridi.setLeafSize(0.02f, 0.02f, 0.02f);
    gridi.setInputCloud(srci);
    EXPECT(!gridi.whyDeferred().empty());  // downsample_all_data_ is the default: the reference averages the intensity
    gridi.setDownsampleAllData(false);      // coordinates only: the device path, the other fields stay default
    EXPECT(gridi.whyDeferred().empty());
    pcl::PointCloud<pcl::PointXYZI> outi;
    gridi.filter(outi);
    bool xyzi = outi.size() == 103 && gridi.deferredReason().empty();
    for (std::size_t i = 0; xyzi && i < outi.size(); ++i) xyzi = outi[i].intensity == 0.0f && outi[i].x == xyz_only[i].x;
    EXPECT(xyzi);
    VoxelGridHIP<pcl::PointXYZRGBNormal> gridc(dev);
    gridc.setInputCloud(load_xyz<pcl::PointXYZRGBNormal>(argv[1]));
    EXPECT(!gridc.whyDeferred().empty());
    gridc.setDownsampleAllData(false);
    EXPECT(gridc.whyDeferred().empty());
  }

  {  // 9. the estimators on explicit pairs through pcl::registration::TransformationEstimation (transformation_estimation.h:74-116)
    pcl::PointCloud<pcl::PointNormal> a, b;
    const float G[16] = {0.9938f, 0.0988f, 0.0517f, 0.1000f, -0.0997f, 0.9949f, 0.0149f, -0.2000f,
                         -0.0500f, -0.0200f, 0.9986f, 0.3000f, 0, 0, 0, 1};
    for (float x = -5.0f; x <= 5.0f; x += 0.5f)
      for (float y = -5.0f; y <= 5.0f; y += 0.5f) {
        pcl::PointNormal p;
        p.x = x; p.y = y; p.z = 0.1f * x * x + 0.2f * x * y - 0.3f * y + 1.0f;
        float nx = -0.2f * x - 0.2f, ny = 0.6f * y - 0.2f, nz = 1.0f;
        const float m = std::sqrt(nx * nx + ny * ny + nz * nz);
        p.normal_x = nx / m; p.normal_y = ny / m; p.normal_z = nz / m;
        a.push_back(p);
      }
    b = a;
    pclhip_transform_cloud(dev->get(), G, 1, a.points.data(), b.points.data(), sizeof(pcl::PointNormal), a.size(), 16);
    using TE = pcl::registration::TransformationEstimation<pcl::PointNormal, pcl::PointNormal, float>;
    const TransformationEstimationSVDHIP<pcl::PointNormal, pcl::PointNormal> svd(dev);
    const TransformationEstimationPointToPlaneLLSHIP<pcl::PointNormal, pcl::PointNormal> lls(dev);
    const TransformationEstimationSymmetricPointToPlaneLLSHIP<pcl::PointNormal, pcl::PointNormal> sym(dev);
    const TE* estimators[3] = {&svd, &lls, &sym};
    const float tol[3] = {1e-4f, 1e-2f, 1e-2f};  // test/registration/test_registration_api.cpp:383-424, :469-518, :663-712
    pcl::Indices all(a.size());
    pcl::Correspondences pairs;
    for (std::size_t i = 0; i < a.size(); ++i) { all[i] = pcl::index_t(i); pairs.emplace_back(pcl::index_t(i), pcl::index_t(i), 0.0f); }
    for (int e = 0; e < 3; ++e) {
      TE::Matrix4 T0 = TE::Matrix4::Identity(), T1 = T0, T2 = T0, T3 = T0;
      estimators[e]->estimateRigidTransformation(a, b, T0);
      estimators[e]->estimateRigidTransformation(a, all, b, T1);
      estimators[e]->estimateRigidTransformation(a, all, b, all, T2);
      estimators[e]->estimateRigidTransformation(a, b, pairs, T3);
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
          EXPECT(std::fabs(T0(r, c) - G[4 * r + c]) < tol[e]);
          EXPECT(T1(r, c) == T0(r, c) && T2(r, c) == T0(r, c) && T3(r, c) == T0(r, c));
        }
    }
  }

  if (failures == 0) std::printf("ALL OK\n");
  return failures == 0 ? 0 : 1;
}
