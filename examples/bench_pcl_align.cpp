// bench_pcl_align -- what the drop-in costs at the boundary a PCL user sees: pcl::Registration::align() on
// IterativeClosestPoint[WithNormals]HIP with HOST pcl::PointClouds (registration/include/pcl/registration/impl/
// registration.hpp:170-221 -> impl/icp.hpp:49-268), next to the iterations themselves.
//
//   bench_pcl_align <target.f32> <source.f32> <points> <mode 0|1>      (raw float32 x,y,z,w records; bench.py writes them)
//
// Built against the PCL mock here (tests/cpp/pcl_mock: PCL's classes with their real signatures, no Eigen); against a
// real PCL tree the same source compiles with -I<pcl>/include.  Prints one JSON object: milliseconds of
//   normals         NormalEstimationHIP::compute on the target (mode 1; device kernel + the Normal cloud back on the host)
//   concatenate     building the PointNormal clouds on the host (mode 1; what pcl::concatenateFields costs, not ours)
//   align_first     the first align(): the target tree built inside initCompute (upload of the records + index build;
//                   PointNormal records bring their normals in the same upload), source upload + ordering, the device loop,
//                   the moved cloud back in `output`;  upload / loop / output are the binding's own split of
//                   computeTransformation, loop_gpu the device time of the iterations
//   align_again     a second align() of the same objects (target and source resident, PCL still copies input -> output)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

#include "pclhip/pcl_plugin.hpp"

using namespace pclhip::plugin;
using Clock = std::chrono::steady_clock;
static double ms(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

template <typename PointT>
static typename pcl::PointCloud<PointT>::Ptr load(const char* path, std::size_t n) {
  typename pcl::PointCloud<PointT>::Ptr c(new pcl::PointCloud<PointT>);
  std::vector<float> raw(n * 4);
  FILE* f = std::fopen(path, "rb");
  if (!f || std::fread(raw.data(), sizeof(float), raw.size(), f) != raw.size()) {
    std::fprintf(stderr, "cannot read %s\n", path);
    std::exit(2);
  }
  std::fclose(f);
  c->points.resize(n);
  for (std::size_t i = 0; i < n; ++i) {
    c->points[i].x = raw[4 * i];
    c->points[i].y = raw[4 * i + 1];
    c->points[i].z = raw[4 * i + 2];
  }
  c->width = std::uint32_t(n);
  c->height = 1;
  c->is_dense = true;
  return c;
}

template <typename Reg, typename Cloud>
static void run_align(Reg& reg, Cloud& out, const char* name) {
  const auto t0 = Clock::now();
  reg.align(out);
  const auto t1 = Clock::now();
  const auto& tm = reg.lastTimings();
  std::printf("\"%s\": {\"total_ms\": %.3f, \"upload_ms\": %.3f, \"loop_ms\": %.3f, \"output_ms\": %.3f, \"iterations\": %d, "
              "\"converged\": %d, \"deferred\": \"%s\"}",
              name, ms(t0, t1), tm.upload_ms, tm.loop_ms, tm.output_ms, reg.iterations(), reg.hasConverged() ? 1 : 0,
              reg.deferredReason().c_str());
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const std::size_t n = std::size_t(std::atoll(argv[3]));
  const int mode = std::atoi(argv[4]);
  auto dev = std::make_shared<Device>(0);
  if (!dev->ok()) {
    std::fprintf(stderr, "no device: %s\n", pclhip_last_error(nullptr));
    return 3;
  }
  auto target = load<pcl::PointXYZ>(argv[1], n);
  auto source = load<pcl::PointXYZ>(argv[2], n);
  std::printf("{\"points\": %zu, \"mode\": %d, ", n, mode);
  if (mode == 0) {
    auto tree = std::make_shared<KdTreeHIP<pcl::PointXYZ>>(dev);  // built by align(): Registration::initCompute
    IterativeClosestPointHIP<pcl::PointXYZ, pcl::PointXYZ> reg(dev);
    reg.setSearchMethodTarget(tree);
    reg.setInputTarget(target);
    reg.setInputSource(source);
    reg.setMaximumIterations(20);
    reg.setMaxCorrespondenceDistance(0.1);
    reg.setTransformationEpsilon(1e-10);
    pcl::PointCloud<pcl::PointXYZ> out;
    run_align(reg, out, "align_first");
    std::printf(", ");
    run_align(reg, out, "align_again");
  } else {
    NormalEstimationHIP<pcl::PointXYZ, pcl::Normal> ne(dev);
    auto t0 = Clock::now();
    ne.setInputCloud(target);
    ne.setKSearch(8);
    ne.setViewPoint(0, 0, 10);
    pcl::PointCloud<pcl::Normal> normals;
    ne.compute(normals);  // builds its own KdTreeHIP over the target first
    auto t1 = Clock::now();
    pcl::PointCloud<pcl::PointNormal>::Ptr tn(new pcl::PointCloud<pcl::PointNormal>), sn(new pcl::PointCloud<pcl::PointNormal>);
    tn->points.resize(n);
    sn->points.resize(n);
    for (std::size_t i = 0; i < n; ++i) {
      auto& p = tn->points[i];
      p.x = (*target)[i].x; p.y = (*target)[i].y; p.z = (*target)[i].z;
      p.normal_x = normals[i].normal_x; p.normal_y = normals[i].normal_y; p.normal_z = normals[i].normal_z;
      p.curvature = normals[i].curvature;
      auto& q = sn->points[i];
      q.x = (*source)[i].x; q.y = (*source)[i].y; q.z = (*source)[i].z;
    }
    tn->width = sn->width = std::uint32_t(n);
    tn->height = sn->height = 1;
    auto t2 = Clock::now();
    std::printf("\"normals_ms\": %.3f, \"concatenate_ms\": %.3f, ", ms(t0, t1), ms(t1, t2));
    auto tree = std::make_shared<KdTreeHIP<pcl::PointNormal>>(dev);  // built by align(): Registration::initCompute
    IterativeClosestPointWithNormalsHIP<pcl::PointNormal, pcl::PointNormal> reg(dev);
    reg.setSearchMethodTarget(tree);
    reg.setInputTarget(tn);
    reg.setInputSource(sn);
    reg.setMaximumIterations(20);
    reg.setMaxCorrespondenceDistance(0.1);
    reg.setTransformationEpsilon(1e-10);
    pcl::PointCloud<pcl::PointNormal> out;
    run_align(reg, out, "align_first");
    std::printf(", ");
    run_align(reg, out, "align_again");
  }
  std::printf("}\n");
  return 0;
}
