// boundary_probe -- where the first NormalEstimationHIP::compute() on a 10M-point HOST cloud spends its 200 ms
// (examples/bench_pcl_align.cpp reports the sum).  g++ -std=c++17 -O2 -Iinclude -Itests/cpp/pcl_mock scratch/boundary_probe.cpp
//   -Lpcl_amd -lpclhip -Wl,-rpath,$PWD/pcl_amd -o /tmp/boundary_probe && /tmp/boundary_probe 10000000
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <random>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include "pclhip/pcl_plugin.hpp"
using namespace pclhip::plugin;
using Clock = std::chrono::steady_clock;
static double ms(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }
int main(int argc, char** argv) {
  const std::size_t n = argc > 1 ? std::size_t(std::atoll(argv[1])) : 10000000;
  auto t0 = Clock::now();
  auto dev = std::make_shared<Device>(0);
  auto t1 = Clock::now();
  if (!dev->ok()) return 3;
  pcl::PointCloud<pcl::PointXYZ>::Ptr cloud(new pcl::PointCloud<pcl::PointXYZ>);
  cloud->points.resize(n);
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  for (auto& p : cloud->points) { p.x = u(rng); p.y = u(rng); p.z = 0.1f * p.x * p.y; }
  cloud->width = std::uint32_t(n); cloud->height = 1;
  auto t2 = Clock::now();
  auto tree = std::make_shared<KdTreeHIP<pcl::PointXYZ>>(dev);
  tree->setInputCloud(cloud);
  auto t3 = Clock::now();
  tree->setInputCloud(cloud);
  auto t4 = Clock::now();
  { pcl::PointCloud<pcl::Normal> tmp; tmp.points.resize(n); auto t5 = Clock::now();
    std::printf("context %.1f ms | first setInputCloud (upload + arena + build) %.1f | second %.1f | resize of 10M pcl::Normal (PCL's own) %.1f\n",
                ms(t0, t1), ms(t2, t3), ms(t3, t4), ms(t4, t5)); }
  NormalEstimationHIP<pcl::PointXYZ, pcl::Normal> ne(dev);
  ne.setInputCloud(cloud); ne.setKSearch(8); ne.setViewPoint(0, 0, 10);
  pcl::PointCloud<pcl::Normal> normals;
  auto t6 = Clock::now();
  ne.compute(normals);
  auto t7 = Clock::now();
  ne.compute(normals);
  auto t8 = Clock::now();
  std::printf("NormalEstimationHIP::compute first %.1f ms | again %.1f ms\n", ms(t6, t7), ms(t7, t8));
  // the same normals through the generic path (staging vector + unpacking loop: any output type) -- every value must agree
  NormalEstimationHIP<pcl::PointXYZ, pcl::PointNormal> ne2(dev);
  ne2.setInputCloud(cloud); ne2.setKSearch(8); ne2.setViewPoint(0, 0, 10);
  pcl::PointCloud<pcl::PointNormal> pn;
  auto t9 = Clock::now();
  ne2.compute(pn);
  auto t10 = Clock::now();
  ne2.compute(pn);
  auto t11 = Clock::now();
  std::size_t bad = 0;
  for (std::size_t i = 0; i < n; ++i)
    if (std::memcmp(normals[i].normal, pn[i].normal, 12) != 0 || std::memcmp(&normals[i].curvature, &pn[i].curvature, 4) != 0) ++bad;
  std::printf("generic path (PointNormal output) first %.1f ms | again %.1f ms | records that differ from the pcl::Normal path: %zu of %zu\n",
              ms(t9, t10), ms(t10, t11), bad, n);
  // CorrespondenceEstimationHIP::determineCorrespondences: 10M pcl::Correspondence structs back on the host
  {
    pcl::PointCloud<pcl::PointXYZ>::Ptr src(new pcl::PointCloud<pcl::PointXYZ>(*cloud));
    for (auto& p : src->points) { p.x += 0.001f; p.z += 0.002f; }
    CorrespondenceEstimationHIP<pcl::PointXYZ, pcl::PointXYZ> ce;
    ce.setSearchMethodTarget(std::make_shared<KdTreeHIP<pcl::PointXYZ>>(dev));
    ce.setInputTarget(cloud);
    ce.setInputSource(src);
    pcl::Correspondences corr;
    auto c0 = Clock::now();
    ce.determineCorrespondences(corr, 0.1);
    auto c1 = Clock::now();
    ce.determineCorrespondences(corr, 0.1);
    auto c2 = Clock::now();
    std::printf("determineCorrespondences first %.1f ms | again %.1f ms | %zu pairs\n", ms(c0, c1), ms(c1, c2), corr.size());
  }
  // the registration: where a first align() on PointNormal host clouds spends its time
  {
    pcl::PointCloud<pcl::PointNormal>::Ptr tn(new pcl::PointCloud<pcl::PointNormal>), sn(new pcl::PointCloud<pcl::PointNormal>);
    tn->points.resize(n); sn->points.resize(n);
    for (std::size_t i = 0; i < n; ++i) {
      auto& p = tn->points[i];
      p.x = (*cloud)[i].x; p.y = (*cloud)[i].y; p.z = (*cloud)[i].z;
      p.normal_x = normals[i].normal_x; p.normal_y = normals[i].normal_y; p.normal_z = normals[i].normal_z; p.curvature = normals[i].curvature;
      auto& q = sn->points[i];
      q.x = p.x + 0.001f; q.y = p.y - 0.0005f; q.z = p.z + 0.002f;
    }
    tn->width = sn->width = std::uint32_t(n); tn->height = sn->height = 1;
    auto tree2 = std::make_shared<KdTreeHIP<pcl::PointNormal>>(dev);
    auto a0 = Clock::now();
    tree2->setInputCloud(tn);
    auto a1 = Clock::now();
    tree2->setInputCloud(tn);
    auto a2 = Clock::now();
    IterativeClosestPointWithNormalsHIP<pcl::PointNormal, pcl::PointNormal> reg(dev);
    reg.setSearchMethodTarget(tree2, true);
    reg.setInputTarget(tn); reg.setInputSource(sn);
    reg.setMaximumIterations(20); reg.setMaxCorrespondenceDistance(0.1); reg.setTransformationEpsilon(1e-10);
    pcl::PointCloud<pcl::PointNormal> out;
    auto a3 = Clock::now();
    reg.align(out);
    auto a4 = Clock::now();
    const auto tm1 = reg.lastTimings();
    reg.align(out);
    auto a5 = Clock::now();
    const auto tm2 = reg.lastTimings();
    std::printf("PointNormal tree (480 MB upload + build) first %.1f ms | again %.1f ms\n", ms(a0, a1), ms(a1, a2));
    std::printf("align() first %.1f ms (binding: upload %.1f, loop %.1f, output %.1f) | again %.1f ms (upload %.1f, loop %.1f, output %.1f) | %d iterations\n",
                ms(a3, a4), tm1.upload_ms, tm1.loop_ms, tm1.output_ms, ms(a4, a5), tm2.upload_ms, tm2.loop_ms, tm2.output_ms, reg.iterations());
  }
  return bad == 0 ? 0 : 1;
}
