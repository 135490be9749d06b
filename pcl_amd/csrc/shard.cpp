// shard.cpp -- target sharding for clouds that are spread over the GPUs of a node.
//
// PCL has no multi-GPU path (gpu/containers/src/initialization.cpp:109 picks one device); the contract is the
// survey's: cut the target into G slabs of equal point count, give every rank its slab plus a HALO wide enough
// that a query owned by the rank finds its true nearest neighbour locally -- or correctly finds none within
// max_correspondence_distance --, and route every source point to the rank whose region holds its CURRENT
// position.  The regions are the cells of a kd partition of space (recursive bisection at order statistics along
// the widest axis, unbounded on the outside), so they tile R^3 with half-open boxes [lo, hi): every finite
// point has exactly one owner.  A query q in region R and a target point p with |q - p| <= d satisfy
// p in R dilated by d per axis, which is the halo rule below.
//
// This file is the host code for clouds in HOST memory (order statistics + a filter over the cloud, once per target):
// it needs no GPU, which is what lets the partition / halo / routing logic be tested in multi-process CPU runs.
// Clouds in DEVICE memory are partitioned and selected where they are (shard_dev.hip: same cuts, same lists).  The per-iteration side
// lives in the search kernel (search.hip: RegionBox) and in the record all-reduce (icp_loop.hip).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "pclhip_internal.hpp"

using namespace pclhip;

namespace {

struct HostCloud {  // xyz of every record, on the host (device clouds are copied once)
  std::vector<float> xyz;
  uint64_t n = 0;
};

pclhip_status fetch_cloud(const void* points, size_t stride, uint64_t n, HostCloud& out) {
  out.n = n;
  out.xyz.resize(size_t(n) * 3);
  if (n == 0) return PCLHIP_OK;
  const char* base = static_cast<const char*>(points);
  std::vector<char> staged;
  if (is_device_pointer(points)) {
    staged.resize(size_t(n) * stride);
    if (hipMemcpy(staged.data(), points, staged.size(), hipMemcpyDeviceToHost) != hipSuccess) {
      set_error(nullptr, "cannot copy the cloud to the host for partitioning");
      return PCLHIP_ERR_HIP;
    }
    base = staged.data();
  }
  const unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      for (uint64_t i = n * t / nt; i < n * (t + 1) / nt; ++i)
        std::memcpy(&out.xyz[size_t(i) * 3], base + size_t(i) * stride, 3 * sizeof(float));
    });
  for (auto& x : th) x.join();
  return PCLHIP_OK;
}

struct Region {
  float lo[3], hi[3];
};

// split `ids` (indices of finite points inside `r`) into `parts` regions of near-equal count
void bisect(const HostCloud& c, std::vector<uint32_t>& ids, size_t begin, size_t end, const Region& r, int parts,
            Region* out) {
  if (parts == 1) {
    *out = r;
    return;
  }
  if (end == begin) {
    // no points to cut (fewer finite points than slabs, or every point of the parent tied at its cut): no cut is
    // invented -- the first part keeps the whole cell, the others get an empty one ([hi, hi) on x owns no point), so
    // the regions still tile space with exactly one owner per point
    out[0] = r;
    for (int p = 1; p < parts; ++p) {
      out[p] = r;
      out[p].lo[0] = r.hi[0];
    }
    return;
  }
  // widest axis of the points themselves (the region may be unbounded)
  float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
  float mx[3] = {-mn[0], -mn[0], -mn[0]};
  for (size_t i = begin; i < end; ++i)
    for (int d = 0; d < 3; ++d) {
      const float v = c.xyz[size_t(ids[i]) * 3 + d];
      mn[d] = std::min(mn[d], v);
      mx[d] = std::max(mx[d], v);
    }
  int axis = 0;
  if (end > begin) {
    float best = mx[0] - mn[0];
    if (mx[1] - mn[1] > best) { best = mx[1] - mn[1]; axis = 1; }
    if (mx[2] - mn[2] > best) axis = 2;
  }
  const int left_parts = parts / 2, right_parts = parts - left_parts;
  const size_t count = end - begin;
  size_t k = begin + count * size_t(left_parts) / size_t(parts);  // order statistic at the cut
  float cut = 0.0f;
  {
    if (k >= end) k = end - 1;
    // (coordinate, index) order: the cut value is the same whatever order the ids arrive in
    auto less = [&](uint32_t a, uint32_t b) {
      const float va = c.xyz[size_t(a) * 3 + axis], vb = c.xyz[size_t(b) * 3 + axis];
      return va < vb || (va == vb && a < b);
    };
    std::nth_element(ids.begin() + long(begin), ids.begin() + long(k), ids.begin() + long(end), less);
    cut = c.xyz[size_t(ids[k]) * 3 + axis] + 0.0f;  // (-0.0 -> +0.0: one spelling of the bound, host and device code)
    // left = strictly below the cut (ties go right, like the kernel's x >= lo && x < hi ownership test)
    k = size_t(std::partition(ids.begin() + long(begin), ids.begin() + long(end),
                              [&](uint32_t a) { return c.xyz[size_t(a) * 3 + axis] < cut; }) - ids.begin());
  }
  Region L = r, R = r;
  L.hi[axis] = cut;
  R.lo[axis] = cut;
  bisect(c, ids, begin, k, L, left_parts, out);
  bisect(c, ids, k, end, R, right_parts, out + left_parts);
}

}  // namespace

extern "C" {

pclhip_status pclhip_partition_slabs(const void* points, size_t stride, uint64_t n, int n_slabs, float* regions) {
  if (!regions || n_slabs < 1 || (n > 0 && !points)) return PCLHIP_ERR_INVALID;
  if (stride < 12 || stride % 4 != 0 || n >= 0x7FFFFFFFull) {
    set_error(nullptr, "stride must be a multiple of 4 and >= 12 bytes, the cloud must fit int32 indices");
    return PCLHIP_ERR_INVALID;
  }
  // shard_dev.hip (8-bit cell labels: up to 254 slabs; more go through the host code below, which fetches the cloud)
  if (n > 0 && is_device_pointer(points) && n_slabs <= 254) return partition_slabs_device(points, stride, n, n_slabs, regions);
  HostCloud c;
  pclhip_status st = fetch_cloud(points, stride, n, c);
  if (st != PCLHIP_OK) return st;
  std::vector<uint32_t> ids;
  ids.reserve(size_t(n));
  for (uint64_t i = 0; i < n; ++i) {
    const float* p = &c.xyz[size_t(i) * 3];
    if (std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2])) ids.push_back(uint32_t(i));
  }
  const float inf = std::numeric_limits<float>::infinity();
  Region all = {{-inf, -inf, -inf}, {inf, inf, inf}};
  std::vector<Region> out(size_t(n_slabs), all);
  bisect(c, ids, 0, ids.size(), all, n_slabs, out.data());
  for (int g = 0; g < n_slabs; ++g) {
    std::memcpy(regions + 6 * g, out[size_t(g)].lo, 3 * sizeof(float));
    std::memcpy(regions + 6 * g + 3, out[size_t(g)].hi, 3 * sizeof(float));
  }
  return PCLHIP_OK;
}

pclhip_status pclhip_select_region(const void* points, size_t stride, uint64_t n, const float region[6], double margin,
                                   int32_t* out_indices, uint64_t capacity, uint64_t* out_count) {
  if (!region || !out_count || (n > 0 && !points) || !(margin >= 0.0)) return PCLHIP_ERR_INVALID;
  if (stride < 12 || stride % 4 != 0 || n >= 0x7FFFFFFFull) {
    set_error(nullptr, "stride must be a multiple of 4 and >= 12 bytes, the cloud must fit int32 indices");
    return PCLHIP_ERR_INVALID;
  }
  // dilated box, rounded outwards: a float d2 that passes the double test d2 <= max_dist^2 may belong to a point a
  // few ulp farther than max_dist; the relative slack covers that and the rounding of the bounds themselves
  float lo[3], hi[3];
  const double m = margin * (1.0 + 1e-5) + 1e-30;
  for (int d = 0; d < 3; ++d) {
    lo[d] = std::nextafterf(float(double(region[d]) - m - 1e-6 * std::fabs(double(region[d]))), -std::numeric_limits<float>::infinity());
    hi[d] = std::nextafterf(float(double(region[3 + d]) + m + 1e-6 * std::fabs(double(region[3 + d]))), std::numeric_limits<float>::infinity());
    if (!std::isfinite(region[d])) lo[d] = region[d];
    if (!std::isfinite(region[3 + d])) hi[d] = region[3 + d];
  }
  if (n > 0 && is_device_pointer(points))  // shard_dev.hip: the cloud stays where it is
    return select_region_device(points, stride, n, lo, hi, out_indices, capacity, out_count);
  HostCloud c;
  pclhip_status st = fetch_cloud(points, stride, n, c);
  if (st != PCLHIP_OK) return st;
  const unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<std::vector<int32_t>> part(nt);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      for (uint64_t i = n * t / nt; i < n * (t + 1) / nt; ++i) {
        const float* p = &c.xyz[size_t(i) * 3];
        // non-finite coordinates fail every comparison: such points are never selected
        if (p[0] >= lo[0] && p[0] <= hi[0] && p[1] >= lo[1] && p[1] <= hi[1] && p[2] >= lo[2] && p[2] <= hi[2])
          part[t].push_back(int32_t(i));
      }
    });
  for (auto& x : th) x.join();
  uint64_t total = 0;
  for (const auto& v : part) total += v.size();
  *out_count = total;
  if (total > capacity || (total > 0 && !out_indices)) {
    set_error(nullptr, "index buffer too small for the selected points");
    return PCLHIP_ERR_OVERFLOW;
  }
  std::vector<int32_t> host;
  int32_t* dst = out_indices;
  const bool dev = total > 0 && is_device_pointer(out_indices);
  if (dev) {
    host.resize(size_t(total));
    dst = host.data();
  }
  uint64_t off = 0;
  for (const auto& v : part) {  // threads own ascending ranges: the list is ascending
    if (!v.empty()) std::memcpy(dst + off, v.data(), v.size() * sizeof(int32_t));
    off += v.size();
  }
  if (dev && hipMemcpy(out_indices, host.data(), size_t(total) * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
    set_error(nullptr, "cannot copy the selected indices to the device");
    return PCLHIP_ERR_HIP;
  }
  return PCLHIP_OK;
}

int pclhip_region_owner(const float* regions, int n_slabs, const float xyz[3]) {
  if (!regions || !xyz) return -1;
  for (int g = 0; g < n_slabs; ++g) {
    const float* r = regions + 6 * g;
    if (xyz[0] >= r[0] && xyz[0] < r[3] && xyz[1] >= r[1] && xyz[1] < r[4] && xyz[2] >= r[2] && xyz[2] < r[5]) return g;
  }
  return -1;  // non-finite points have no owner
}

pclhip_status pclhip_icp_set_region(pclhip_icp* icp, const float region[6]) {
  if (!icp) return PCLHIP_ERR_INVALID;
  if (!region) {
    icp->region.on = 0;
    return PCLHIP_OK;
  }
  for (int d = 0; d < 3; ++d) {
    if (!(region[d] <= region[3 + d])) {
      set_error(icp->ctx, "region: lo must not exceed hi");
      return PCLHIP_ERR_INVALID;
    }
    icp->region.lo[d] = region[d];
    icp->region.hi[d] = region[3 + d];
  }
  icp->region.on = 1;
  return PCLHIP_OK;
}

}  // extern "C"
