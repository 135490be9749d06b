// C++ host-side test of the PCL-compatible adapters (include/pclhip/pcl_compat.hpp) over the C ABI.
// Mirrors test/registration/test_registration_api.cpp:83-104 (397 bunny correspondences),
// test/registration/test_registration.cpp:236-270 (ICP golden 4x4 @1e-3) and
// test/filters/test_filters.cpp:566-596 (VoxelGrid 103).  Inputs: bun0.txt bun4.txt golden_corr.txt
// written by the pytest wrapper (tests/test_gpu_cpp_adapters.py) from tests/golden/.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "pclhip/pcl_compat.hpp"

using namespace pclhip;

static PointCloud<PointXYZ>::Ptr load_xyz(const char* path) {
  auto c = std::make_shared<PointCloud<PointXYZ>>();
  std::ifstream f(path);
  float x, y, z;
  while (f >> x >> y >> z) c->push_back(PointXYZ(x, y, z));
  return c;
}

#define EXPECT(cond)                                                        \
  do {                                                                      \
    if (!(cond)) {                                                          \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      ++failures;                                                           \
    }                                                                       \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  int failures = 0;
  auto ctx = std::make_shared<Context>(0);
  if (!ctx->ok()) {
    std::fprintf(stderr, "no device: %s\n", ctx->getLastError().c_str());
    return 3;
  }
  auto source = load_xyz(argv[1]);
  auto target = load_xyz(argv[2]);
  EXPECT(source->size() == 397 && target->size() == 361);

  {  // CorrespondenceEstimation: 397 exact pairs
    std::vector<int> gold;
    std::ifstream f(argv[3]);
    int a, b;
    while (f >> a >> b) gold.push_back(b);
    registration::CorrespondenceEstimation<PointXYZ, PointXYZ> ce(ctx);
    ce.setInputSource(source);
    ce.setInputTarget(target);
    Correspondences corr;
    ce.determineCorrespondences(corr);
    EXPECT(corr.size() == 397 && gold.size() == 397);
    for (std::size_t i = 0; i < corr.size() && i < gold.size(); ++i) {
      EXPECT(corr[i].index_query == int(i));
      EXPECT(corr[i].index_match == gold[i]);
    }
  }
  {  // per-point nearestKSearch through the search::KdTree surface
    auto tree = std::make_shared<search::KdTree<PointXYZ>>(ctx);
    EXPECT(tree->setInputCloud(target));
    Indices idx;
    std::vector<float> d2;
    EXPECT(tree->nearestKSearch((*source)[0], 5, idx, d2) == 5);
    EXPECT(idx.size() == 5 && d2[0] <= d2[1] && d2[1] <= d2[4]);
    std::vector<Indices> bi;
    std::vector<std::vector<float>> bd;
    tree->nearestKSearch(*source, Indices(), 5, bi, bd);
    EXPECT(bi.size() == 397 && bi[0] == idx && bd[0] == d2);
    EXPECT(tree->nearestKSearch((*source)[0], 1000, idx, d2) == 361);  // k clamped to the cloud size
  }
  {  // IterativeClosestPoint golden
    IterativeClosestPoint<PointXYZ, PointXYZ> reg(ctx);
    reg.setInputSource(source);
    reg.setInputTarget(target);
    reg.setMaximumIterations(50);
    reg.setTransformationEpsilon(1e-8);
    reg.setMaxCorrespondenceDistance(0.05);
    PointCloud<PointXYZ> out;
    reg.align(out);
    EXPECT(out.size() == source->size());
    const Matrix4f T = reg.getFinalTransformation();
    const float g[3][4] = {{0.8806f, 0.036481287f, -0.4724f, 0.03453f},
                           {-0.02354f, 0.9992f, 0.03326f, -0.001519f},
                           {0.4732f, -0.01817f, 0.8808f, 0.04116f}};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) EXPECT(std::fabs(T(r, c) - g[r][c]) < ((r == 0 && c == 1) ? 1e-2f : 1e-3f));
    EXPECT(T(3, 0) == 0 && T(3, 1) == 0 && T(3, 2) == 0 && T(3, 3) == 1);
    EXPECT(reg.hasConverged());
  }
  {  // NormalEstimation(k=10) + IterativeClosestPointWithNormals on PointNormal clouds
    NormalEstimation<PointXYZ> ne(ctx);
    ne.setInputCloud(target);
    ne.setKSearch(10);
    PointCloud<Normal> normals;
    ne.compute(normals);
    EXPECT(normals.size() == target->size() && normals.is_dense);
    auto tgt_n = std::make_shared<PointCloud<PointNormal>>();
    auto src_n = std::make_shared<PointCloud<PointNormal>>();
    for (std::size_t i = 0; i < target->size(); ++i) {
      PointNormal p;
      p.x = (*target)[i].x; p.y = (*target)[i].y; p.z = (*target)[i].z;
      p.normal_x = normals[i].normal_x; p.normal_y = normals[i].normal_y; p.normal_z = normals[i].normal_z;
      tgt_n->push_back(p);
    }
    for (std::size_t i = 0; i < source->size(); ++i) {
      PointNormal p;
      p.x = (*source)[i].x; p.y = (*source)[i].y; p.z = (*source)[i].z;
      src_n->push_back(p);
    }
    IterativeClosestPointWithNormals<PointNormal, PointNormal> reg(ctx);
    reg.setInputSource(src_n);
    reg.setInputTarget(tgt_n);
    reg.setMaximumIterations(50);
    reg.setTransformationEpsilon(1e-8);
    PointCloud<PointNormal> out;
    reg.align(out);
    EXPECT(reg.hasConverged());
    EXPECT(reg.getLastMSE() < 1e-3);  // fitness bar of test_registration.cpp:272-318
  }
  {  // reciprocal correspondences (53 pairs) + radiusSearch + a rejector inside ICP
    registration::CorrespondenceEstimation<PointXYZ, PointXYZ> ce(ctx);
    ce.setInputSource(source);
    ce.setInputTarget(target);
    Correspondences corr;
    ce.determineReciprocalCorrespondences(corr);
    EXPECT(corr.size() == 53);
    auto tree = std::make_shared<search::KdTree<PointXYZ>>(ctx);
    EXPECT(tree->setInputCloud(target));
    Indices idx;
    std::vector<float> d2;
    const int n = tree->radiusSearch((*target)[0], 0.01, idx, d2);
    EXPECT(n >= 1 && idx[0] == 0 && d2[0] == 0.0f);
    for (int i = 1; i < n; ++i) EXPECT(d2[i] >= d2[i - 1] && d2[i] < 1e-4f);
    EXPECT(tree->radiusSearch((*target)[0], 0.01, idx, d2, 2) == (n < 2 ? n : 2));
    IterativeClosestPoint<PointXYZ, PointXYZ> reg(ctx);
    reg.setInputSource(source);
    reg.setInputTarget(target);
    reg.setMaximumIterations(30);
    reg.setMaxCorrespondenceDistance(0.05);
    auto rej = std::make_shared<registration::CorrespondenceRejectorMedianDistance>();
    rej->setMedianFactor(4.0);
    reg.addCorrespondenceRejector(rej);
    reg.addCorrespondenceRejector(std::make_shared<registration::CorrespondenceRejectorOneToOne>());
    PointCloud<PointXYZ> out;
    reg.align(out);
    EXPECT(reg.hasConverged() && reg.getNumberOfIterations() > 1);
  }
  {  // VoxelGrid 103
    VoxelGrid<PointXYZ> grid(ctx);
    grid.setLeafSize(0.02f, 0.02f, 0.02f);
    grid.setInputCloud(source);
    PointCloud<PointXYZ> out;
    grid.filter(out);
    EXPECT(out.size() == 103 && out.width == 103 && out.height == 1 && out.is_dense);
    // setSaveLeafLayout (voxel_grid.h:316): every kept cell maps to its centroid's position, in ascending cell order
    grid.setSaveLeafLayout(true);
    grid.filter(out);
    const std::vector<int> layout = grid.getLeafLayout();
    const auto div = grid.getNrDivisions();
    EXPECT(layout.size() == std::size_t(div[0]) * std::size_t(div[1]) * std::size_t(div[2]));
    int next = 0;
    bool ascending = true;
    for (int v : layout)
      if (v >= 0) ascending = ascending && (v == next++);
    EXPECT(ascending && next == 103);
    EXPECT(grid.getCentroidIndex((*source)[0]) >= 0 && grid.getCentroidIndex((*source)[0]) < 103);
    const auto mn = grid.getMinBoxCoordinates();
    EXPECT(grid.getCentroidIndexAt({mn[0] - 1, mn[1], mn[2]}) == -1);
  }
  {  // VoxelGrid<PointNormal>: all fields averaged (default) against coordinates only (voxel_grid.h:293-301)
    auto cloud = std::make_shared<PointCloud<PointNormal>>();
    for (std::size_t i = 0; i < source->size(); ++i) {
      PointNormal p;
      p.x = (*source)[i].x; p.y = (*source)[i].y; p.z = (*source)[i].z;
      p.normal_x = 0.0f; p.normal_y = 0.0f; p.normal_z = 1.0f; p.curvature = 0.25f;
      cloud->push_back(p);
    }
    VoxelGrid<PointNormal> grid(ctx);
    grid.setLeafSize(0.02f, 0.02f, 0.02f);
    grid.setInputCloud(cloud);
    EXPECT(grid.getDownsampleAllData() && grid.getLeafSize()[1] == 0.02f && grid.getMinimumPointsNumberPerVoxel() == 0);
    PointCloud<PointNormal> all, xyz_only;
    grid.filter(all);
    grid.setDownsampleAllData(false);
    grid.filter(xyz_only);
    EXPECT(all.size() == 103 && xyz_only.size() == 103);
    bool same_xyz = true, normals_ok = true, defaults_ok = true;
    for (std::size_t i = 0; i < all.size() && i < xyz_only.size(); ++i) {
      same_xyz = same_xyz && all[i].x == xyz_only[i].x && all[i].y == xyz_only[i].y && all[i].z == xyz_only[i].z;
      normals_ok = normals_ok && all[i].normal_z == 1.0f && all[i].normal_x == 0.0f && all[i].curvature == 0.25f;
      defaults_ok = defaults_ok && xyz_only[i].normal_z == 0.0f && xyz_only[i].curvature == 0.0f;
    }
    EXPECT(same_xyz && normals_ok && defaults_ok);
    grid.setFilterFieldName("intensity");  // not a field of pcl::PointNormal: refused, not ignored
    grid.filter(all);
    EXPECT(all.size() == 0);
    // the pass-through filter on another field, and its negative: the two halves partition the cloud's voxels' points
    grid.setDownsampleAllData(true);
    grid.setFilterFieldName("y");
    grid.setFilterLimits(0.10, 1.0);
    PointCloud<PointNormal> upper, lower;
    grid.filter(upper);
    grid.setFilterLimitsNegative(true);
    EXPECT(grid.getFilterLimitsNegative());
    grid.filter(lower);
    bool split = upper.size() > 0 && lower.size() > 0;
    for (const auto& p : upper.points) split = split && p.y >= 0.10f;
    for (const auto& p : lower.points) split = split && p.y <= 0.10f;
    EXPECT(split);
  }
  {  // NormalEstimation with a search surface and an index subset (feature.h:139-153, pcl_base.h:102-125)
    auto tree = std::make_shared<search::KdTree<PointXYZ>>(ctx);
    NormalEstimation<PointXYZ> ne(ctx);
    ne.setSearchMethod(tree);
    ne.setKSearch(10);
    ne.setViewPoint(0.0f, 0.0f, 10.0f);
    ne.setInputCloud(target);
    PointCloud<Normal> self, same, sub, cross;
    ne.compute(self);
    EXPECT(self.size() == target->size());
    ne.setSearchSurface(target);  // the input itself as an explicit surface: the same normals through the query path
    auto every_third = std::make_shared<Indices>();
    for (std::size_t i = 0; i < target->size(); i += 3) every_third->push_back(index_t(i));
    ne.setIndices(every_third);
    ne.compute(sub);
    EXPECT(sub.size() == every_third->size());
    bool equal = true;
    for (std::size_t j = 0; j < sub.size(); ++j) {
      const Normal& a = sub[j];
      const Normal& b = self[std::size_t((*every_third)[j])];
      equal = equal && a.normal_x == b.normal_x && a.normal_y == b.normal_y && a.normal_z == b.normal_z && a.curvature == b.curvature;
    }
    EXPECT(equal);
    NormalEstimationOMP<PointXYZ> omp(ctx, 8);  // normal_3d_omp.h: the same normals whatever the thread count
    omp.setSearchMethod(tree);
    omp.setKSearch(10);
    omp.setViewPoint(0.0f, 0.0f, 10.0f);
    omp.setInputCloud(target);
    omp.compute(same);
    bool omp_equal = same.size() == self.size() && omp.getNumberOfThreads() == 8;
    for (std::size_t j = 0; omp_equal && j < same.size(); ++j)
      omp_equal = same[j].normal_x == self[j].normal_x && same[j].normal_z == self[j].normal_z && same[j].curvature == self[j].curvature;
    EXPECT(omp_equal);
    NormalEstimation<PointXYZ> ne2(ctx);  // normals AT the source points from the target surface
    ne2.setKSearch(10);
    ne2.setViewPoint(0.0f, 0.0f, 10.0f);
    ne2.setInputCloud(source);
    ne2.setSearchSurface(target);
    ne2.compute(cross);
    EXPECT(cross.size() == source->size() && ne2.getSearchSurface() == target && ne2.getKSearch() == 10);
    bool unit = true;
    for (std::size_t j = 0; j < cross.size(); ++j) {
      const float l = cross[j].normal_x * cross[j].normal_x + cross[j].normal_y * cross[j].normal_y + cross[j].normal_z * cross[j].normal_z;
      unit = unit && std::fabs(l - 1.0f) < 1e-4f;
    }
    EXPECT(unit);
  }
  {  // the criteria object of the reference (icp.h:180-184) and the stored-only options of Registration / KdTree
    IterativeClosestPoint<PointXYZ, PointXYZ> reg(ctx);
    reg.setInputSource(source);
    reg.setInputTarget(target);
    reg.setMaximumIterations(3);
    reg.getConvergeCriteria()->setFailureAfterMaximumIterations(true);
    reg.setRANSACIterations(5);
    reg.setRANSACOutlierRejectionThreshold(0.1);
    EXPECT(reg.getRANSACIterations() == 5 && reg.getRANSACOutlierRejectionThreshold() == 0.1);
    PointCloud<PointXYZ> out;
    reg.align(out);
    // three iterations do not converge the bunny pair: with the flag set that is a failure, not a convergence
    EXPECT(!reg.hasConverged());
    EXPECT(reg.getConvergeCriteria()->getConvergenceState() ==
           registration::DefaultConvergenceCriteria::CONVERGENCE_CRITERIA_FAILURE_AFTER_MAX_ITERATIONS);
    EXPECT(reg.getConvergeCriteria()->getMaximumIterations() == 3);
    auto tree = std::make_shared<search::KdTree<PointXYZ>>(ctx);
    tree->setEpsilon(0.1f);
    tree->setMinPts(3);
    EXPECT(tree->getEpsilon() == 0.1f && tree->getMinPts() == 3);
    EXPECT(tree->setInputCloud(target));
    Indices idx;
    std::vector<float> d2;
    PointNormal q;
    q.x = (*target)[5].x; q.y = (*target)[5].y; q.z = (*target)[5].z;
    EXPECT(tree->nearestKSearchT(q, 1, idx, d2) == 1 && idx[0] == 5 && d2[0] == 0.0f);
    registration::CorrespondenceEstimation<PointXYZ, PointXYZ> ce(ctx);
    EXPECT(!ce.requiresSourceNormals() && !ce.requiresTargetNormals());
    ce.setNumberOfThreads(8);
    auto src_tree = std::make_shared<search::KdTree<PointXYZ>>(ctx);
    ce.setSearchMethodSource(src_tree);
    EXPECT(ce.getSearchMethodSource() == src_tree);
  }
  {  // estimators on explicit pairs: test/registration/test_registration_api.cpp:469-518, :663-712 (1e-2)
    PointCloud<PointNormal> src, tgt;
    const float G[16] = {0.9938f, 0.0988f, 0.0517f, 0.1000f, -0.0997f, 0.9949f, 0.0149f, -0.2000f,
                         -0.0500f, -0.0200f, 0.9986f, 0.3000f, 0, 0, 0, 1};
    for (float x = -5.0f; x <= 5.0f; x += 0.5f)
      for (float y = -5.0f; y <= 5.0f; y += 0.5f) {
        PointNormal p;
        p.x = x; p.y = y; p.z = 0.1f * x * x + 0.2f * x * y - 0.3f * y + 1.0f;
        float nx = -0.2f * x - 0.2f, ny = 0.6f * y - 0.2f, nz = 1.0f;
        const float m = std::sqrt(nx * nx + ny * ny + nz * nz);
        p.normal_x = nx / m; p.normal_y = ny / m; p.normal_z = nz / m;
        src.push_back(p);
      }
    tgt = src;
    pclhip_transform_cloud(ctx->get(), G, 1, src.points.data(), tgt.points.data(), sizeof(PointNormal), src.size(), 16);
    Matrix4f T;
    registration::TransformationEstimationSymmetricPointToPlaneLLS<PointNormal, PointNormal> sym(ctx);
    sym.estimateRigidTransformation(src, tgt, T);
    for (int i = 0; i < 16; ++i) EXPECT(std::fabs(T.m[i] - G[i]) < 1e-2f);
    registration::TransformationEstimationPointToPlaneLLS<PointNormal, PointNormal> lls(ctx);
    T = Matrix4f();
    lls.estimateRigidTransformation(src, tgt, T);
    for (int i = 0; i < 16; ++i) EXPECT(std::fabs(T.m[i] - G[i]) < 1e-2f);
    registration::TransformationEstimationSVD<PointNormal, PointNormal> svd(ctx);
    T = Matrix4f();
    svd.estimateRigidTransformation(src, tgt, T);
    for (int i = 0; i < 16; ++i) EXPECT(std::fabs(T.m[i] - G[i]) < 1e-4f);
    // the other three overloads of transformation_estimation.h:74-116, through the abstract base
    const registration::TransformationEstimation<PointNormal, PointNormal>& base = svd;
    Indices all(src.size());
    for (std::size_t i = 0; i < all.size(); ++i) all[i] = index_t(i);
    Correspondences pairs;
    for (std::size_t i = 0; i < src.size(); ++i) pairs.emplace_back(index_t(i), index_t(i), 0.0f);
    Matrix4f Ta, Tb, Tc;
    base.estimateRigidTransformation(src, all, tgt, Ta);
    base.estimateRigidTransformation(src, all, tgt, all, Tb);
    base.estimateRigidTransformation(src, tgt, pairs, Tc);
    for (int i = 0; i < 16; ++i) EXPECT(Ta.m[i] == T.m[i] && Tb.m[i] == T.m[i] && Tc.m[i] == T.m[i]);
    Matrix4f untouched;
    untouched(0, 3) = 42.0f;
    PointCloud<PointNormal> shorter = tgt;
    shorter.points.pop_back();
    base.estimateRigidTransformation(src, shorter, untouched);  // size mismatch: the matrix is left alone
    EXPECT(untouched(0, 3) == 42.0f);
    // IterativeClosestPointWithNormals + setUseSymmetricObjective on PointNormal clouds: a motion small
    // against the 0.5 grid spacing, so nearest neighbours are the true pairs and the motion is recovered
    const float c = std::cos(0.02f), sn = std::sin(0.02f);
    const float G2[16] = {c, -sn, 0, 0.03f, sn, c, 0, -0.02f, 0, 0, 1, 0.04f, 0, 0, 0, 1};
    PointCloud<PointNormal> tgt_small = src;
    pclhip_transform_cloud(ctx->get(), G2, 1, src.points.data(), tgt_small.points.data(), sizeof(PointNormal), src.size(), 16);
    auto s2 = std::make_shared<PointCloud<PointNormal>>(src);
    auto t2 = std::make_shared<PointCloud<PointNormal>>(tgt_small);
    IterativeClosestPointWithNormals<PointNormal, PointNormal> reg(ctx);
    reg.setUseSymmetricObjective(true);
    EXPECT(reg.getUseSymmetricObjective());
    reg.setInputSource(s2);
    reg.setInputTarget(t2);
    reg.setMaximumIterations(50);
    PointCloud<PointNormal> out;
    reg.align(out);
    EXPECT(reg.hasConverged());
    const Matrix4f F = reg.getFinalTransformation();
    for (int i = 0; i < 16; ++i) EXPECT(std::fabs(F.m[i] - G2[i]) < 1e-3f);
    EXPECT(reg.getFitnessScore() < 1e-6);
  }
  {  // PCD round trip through pcl::io-style calls (binary_compressed, PointNormal and PointXYZ)
    const std::string dir = std::string(argv[1]).substr(0, std::string(argv[1]).find_last_of('/') + 1);
    PointCloud<PointNormal> pn;
    for (int i = 0; i < 500; ++i) {
      PointNormal p;
      p.x = 0.01f * float(i); p.y = float(i % 7); p.z = -1.5f;
      p.normal_x = 0; p.normal_y = 0.6f; p.normal_z = 0.8f; p.curvature = 0.001f * float(i);
      pn.push_back(p);
    }
    EXPECT(io::savePCDFileBinaryCompressed(dir + "pn.pcd", pn) == 0);
    PointCloud<PointNormal> back;
    EXPECT(io::loadPCDFile(dir + "pn.pcd", back) == 0);
    EXPECT(back.size() == 500 && back.width == 500 && back.height == 1 && back.is_dense);
    for (std::size_t i = 0; i < back.size() && i < pn.size(); ++i)
      EXPECT(back[i].x == pn[i].x && back[i].y == pn[i].y && back[i].z == pn[i].z && back[i].normal_y == 0.6f &&
             back[i].normal_z == 0.8f && back[i].curvature == pn[i].curvature);
    EXPECT(io::savePCDFileASCII(dir + "xyz.pcd", *source) == 0);
    PointCloud<PointXYZ> sb;
    EXPECT(io::loadPCDFile(dir + "xyz.pcd", sb) == 0);
    EXPECT(sb.size() == source->size());
    for (std::size_t i = 0; i < sb.size() && i < source->size(); ++i)
      EXPECT(std::fabs(sb[i].x - (*source)[i].x) <= 1e-7f * std::fabs((*source)[i].x) && sb[i].w == 1.0f);
    EXPECT(io::loadPCDFile(dir + "does_not_exist.pcd", sb) == -1);
  }
  {  // the plug structure: default-constructible objects, abstract bases, foreign estimators through virtual calls
    // a foreign TransformationEstimation (registration.h:144-148): wraps the stock SVD and counts its calls
    struct CountingSVD : registration::TransformationEstimation<PointXYZ, PointXYZ> {
      registration::TransformationEstimationSVD<PointXYZ, PointXYZ> inner;
      mutable int calls = 0;
      void estimateRigidTransformation(const PointCloud<PointXYZ>& a, const PointCloud<PointXYZ>& b, Matrix4f& T) const override {
        inner.estimateRigidTransformation(a, b, T);
      }
      void estimateRigidTransformation(const PointCloud<PointXYZ>& a, const Indices& ia, const PointCloud<PointXYZ>& b,
                                       Matrix4f& T) const override { inner.estimateRigidTransformation(a, ia, b, T); }
      void estimateRigidTransformation(const PointCloud<PointXYZ>& a, const Indices& ia, const PointCloud<PointXYZ>& b,
                                       const Indices& ib, Matrix4f& T) const override { inner.estimateRigidTransformation(a, ia, b, ib, T); }
      void estimateRigidTransformation(const PointCloud<PointXYZ>& a, const PointCloud<PointXYZ>& b, const Correspondences& c,
                                       Matrix4f& T) const override {
        ++calls;
        inner.estimateRigidTransformation(a, b, c, T);
      }
    };
    // a foreign CorrespondenceEstimation (registration.h:173-177): the batch search of a Search object it was given
    struct BatchCE : registration::CorrespondenceEstimationBase<PointXYZ, PointXYZ> {
      int calls = 0;
      void determineCorrespondences(Correspondences& out, double max_distance) override {
        out.clear();
        ++calls;
        if (!this->initCompute()) return;
        const search::Search<PointXYZ>& s = *this->tree_;  // only the abstract interface is used
        std::vector<Indices> ki;
        std::vector<std::vector<float>> kd;
        s.nearestKSearch(*this->input_, Indices(), 1, ki, kd);
        for (std::size_t i = 0; i < ki.size(); ++i)
          if (!ki[i].empty() && double(kd[i][0]) <= max_distance * max_distance) out.emplace_back(index_t(i), ki[i][0], kd[i][0]);
      }
      void determineReciprocalCorrespondences(Correspondences& out, double d) override { determineCorrespondences(out, d); }
      Ptr clone() const override { return std::make_shared<BatchCE>(*this); }
    };
    IterativeClosestPoint<PointXYZ, PointXYZ> stock;  // default constructor: the default context
    stock.setInputSource(source);
    stock.setInputTarget(target);
    stock.setMaximumIterations(50);
    stock.setTransformationEpsilon(1e-8);
    stock.setMaxCorrespondenceDistance(0.05);
    PointCloud<PointXYZ> out;
    stock.align(out);
    EXPECT(stock.hasConverged() && stock.ranOnDeviceLoop());
    const Matrix4f Ts = stock.getFinalTransformation();
    const Matrix4f last = stock.getLastIncrementalTransformation();
    EXPECT(std::fabs(last(0, 0) - 1.0f) < 1e-3f && std::fabs(last(0, 3)) < 1e-3f);  // the last step was tiny

    Registration<PointXYZ, PointXYZ>* reg = nullptr;
    IterativeClosestPoint<PointXYZ, PointXYZ> plugged;
    reg = &plugged;                                         // driven through the abstract Registration
    auto te = std::make_shared<CountingSVD>();
    auto ce = std::make_shared<BatchCE>();
    reg->setTransformationEstimation(te);
    reg->setCorrespondenceEstimation(ce);
    reg->setInputSource(source);
    reg->setInputTarget(target);
    reg->setMaximumIterations(50);
    reg->setTransformationEpsilon(1e-8);
    reg->setMaxCorrespondenceDistance(0.05);
    reg->align(out);
    EXPECT(reg->hasConverged() && !plugged.ranOnDeviceLoop());
    EXPECT(te->calls == plugged.getNumberOfIterations() && ce->calls == te->calls && te->calls > 3);
    EXPECT(plugged.getNumberOfIterations() == stock.getNumberOfIterations());
    const Matrix4f Tp = reg->getFinalTransformation();
    for (int i = 0; i < 16; ++i) EXPECT(std::fabs(Tp.m[i] - Ts.m[i]) < 2e-5f);  // same loop, host-side sums order aside
    EXPECT(out.size() == source->size());

    // PCLBase::setIndices on the registration: only those source points take part
    auto half = std::make_shared<Indices>();
    for (int i = 0; i < 397; i += 2) half->push_back(i);
    stock.setIndices(half);
    stock.align(out);
    EXPECT(stock.hasConverged() && out.size() == source->size());
    const Matrix4f Th = stock.getFinalTransformation();
    for (int r = 0; r < 3; ++r) EXPECT(std::fabs(Th(r, 3) - Ts(r, 3)) < 3e-3f);

    // a search method used through the abstract base + point representations (kdtree.h:110)
    search::Search<PointXYZ>::Ptr s = std::make_shared<search::KdTree<PointXYZ>>();
    EXPECT(s->setInputCloud(target) && s->getName() == "KdTree");
    Indices idx;
    std::vector<float> d2;
    EXPECT(s->nearestKSearch(*source, 7, 3, idx, d2) == 3);
    auto tree = std::make_shared<search::KdTree<PointXYZ>>();
    auto xy = std::make_shared<CustomPointRepresentation<PointXYZ>>(2, 0);
    tree->setPointRepresentation(xy);
    EXPECT(tree->setInputCloud(target));
    PointXYZ q = (*target)[17];
    q.z += 5.0f;
    EXPECT(tree->nearestKSearch(q, 1, idx, d2) == 1 && idx[0] == 17 && d2[0] == 0.0f);
    auto weird = std::make_shared<CustomPointRepresentation<PointXYZ>>(3, 1);  // (y, z, w): not a rescaled x y z
    tree->setPointRepresentation(weird);
    EXPECT(!tree->setInputCloud(target));                                        // refused, not silently different

    // CorrespondenceEstimationBase: clone() and source / target index subsets
    registration::CorrespondenceEstimationBase<PointXYZ, PointXYZ>::Ptr base = std::make_shared<registration::CorrespondenceEstimation<PointXYZ, PointXYZ>>();
    base->setInputSource(source);
    base->setInputTarget(target);
    auto some = std::make_shared<Indices>(Indices{3, 10, 200, 396});
    base->setIndicesSource(some);
    Correspondences corr;
    base->determineCorrespondences(corr);
    EXPECT(corr.size() == 4 && corr[0].index_query == 3 && corr[3].index_query == 396);
    auto copy = base->clone();
    Correspondences corr2;
    copy->determineCorrespondences(corr2);
    EXPECT(corr2.size() == 4 && corr2[2].index_match == corr[2].index_match);
    auto evens = std::make_shared<Indices>();
    for (int i = 0; i < 361; i += 2) evens->push_back(i);
    base->setIndicesTarget(evens);
    base->determineCorrespondences(corr);
    for (const auto& c : corr) EXPECT(c.index_match % 2 == 0);
  }
  std::printf(failures ? "%d FAILURES\n" : "ALL OK\n", failures);
  return failures ? 1 : 0;
}
