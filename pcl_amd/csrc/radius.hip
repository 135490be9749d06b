// radius.hip -- batched radiusSearch on the same index.
// Replaces pcl::KdTreeFLANN<PointT>::radiusSearch (kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp
// :372-414): all indexed points with squared distance < float(radius*radius) (FLANN's
// RadiusResultSet keeps `dist < radius`), ascending by (distance, index), optionally only the
// max_nn nearest.  Two traversals (count, then fill at exclusive-scan offsets) + one segmented
// radix sort of (distance, index) keys.
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include <cfloat>
#include <cmath>

#include "traverse.hpp"

namespace pclhip {
namespace {

constexpr int BLOCK = 256;
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;

struct RadiusCount {
  static constexpr int QPL = 1;
  float r2;
  uint32_t cnt;
  __device__ __forceinline__ float worst(int) const { return r2; }
  __device__ __forceinline__ void leaf(const float* l, uint32_t, const float* qx, const float* qy, const float* qz) {
    const v2f qx2 = {qx[0], qx[0]}, qy2 = {qy[0], qy[0]}, qz2 = {qz[0], qz[0]};
#pragma unroll
    for (int j = 0; j < LEAF / 2; ++j) {
      const v2f r = pair_dist(l, j, qx2, qy2, qz2);
      cnt += (r.x < r2 ? 1u : 0u) + (r.y < r2 ? 1u : 0u);
    }
  }
};

struct RadiusFill {
  static constexpr int QPL = 1;
  float r2;
  uint64_t* out;  // this query's segment
  uint32_t cnt;
  bool active;
  __device__ __forceinline__ float worst(int) const { return r2; }
  __device__ __forceinline__ void leaf(const float* l, uint32_t, const float* qx, const float* qy, const float* qz) {
    const v2f qx2 = {qx[0], qx[0]}, qy2 = {qy[0], qy[0]}, qz2 = {qz[0], qz[0]};
#pragma unroll
    for (int j = 0; j < LEAF / 2; ++j) {
      const v2f r = pair_dist(l, j, qx2, qy2, qz2);
      if (active && r.x < r2) out[cnt++] = make_key(r.x, __float_as_uint(l[3 * LEAF + 2 * j]));
      if (active && r.y < r2) out[cnt++] = make_key(r.y, __float_as_uint(l[3 * LEAF + 2 * j + 1]));
    }
  }
};

template <bool FILL>
__global__ __launch_bounds__(BLOCK) void radius_kernel(IndexView ix, const float4* __restrict__ q, uint32_t nq, float r2,
                                                       uint32_t* __restrict__ counts,
                                                       const unsigned long long* __restrict__ offsets,
                                                       uint64_t* __restrict__ keys) {
  __shared__ WaveLds wl_s[WAVES_PER_BLOCK];
  __shared__ Box topbox_s[TOPCACHE_BOXES];
  load_top_cache(ix, topbox_s);
  const int lane = threadIdx.x & (WAVE - 1);
  const uint32_t ngroups = (nq + WAVE - 1) / WAVE;
  const GroupSchedule sched(ngroups);
  TraverseStats ts;
  for (uint32_t gl = sched.first(); gl < sched.groups_per_xcd; gl += sched.step()) {
    const uint32_t g = sched.global(gl);
    if (g >= ngroups) break;
    const uint32_t i = g * WAVE + lane;
    float4 p = make_float4(0, 0, 0, 0);
    const bool real = i < nq;
    if (real) p = q[i];
    const uint32_t oq = __float_as_uint(p.w);
    const bool vv[1] = {real && isfinite(p.x) && isfinite(p.y) && isfinite(p.z)};
    const float qx[1] = {p.x}, qy[1] = {p.y}, qz[1] = {p.z};
    if constexpr (!FILL) {
      RadiusCount pol;
      pol.r2 = r2;
      pol.cnt = 0;
      traverse(ix, qx, qy, qz, vv, pol, wl_s[threadIdx.x / WAVE], topbox_s, ts);
      if (real) counts[oq] = vv[0] ? pol.cnt : 0u;
    } else {
      RadiusFill pol;
      pol.r2 = r2;
      pol.cnt = 0;
      pol.active = vv[0];
      pol.out = keys + (real ? offsets[oq] : 0ull);
      traverse(ix, qx, qy, qz, vv, pol, wl_s[threadIdx.x / WAVE], topbox_s, ts);
    }
  }
}

__global__ void clamp_counts_kernel(const uint32_t* counts, uint32_t nq, uint32_t max_nn, unsigned long long* clamped) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) clamped[i] = (max_nn && counts[i] > max_nn) ? max_nn : counts[i];
}
__global__ void widen_counts_kernel(const uint32_t* counts, uint32_t nq, unsigned long long* wide) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) wide[i] = counts[i];
}

// sorted full segments -> (possibly truncated) output CSR
__global__ void radius_emit_kernel(const uint64_t* __restrict__ keys, const unsigned long long* __restrict__ full_off,
                                   const unsigned long long* __restrict__ out_off, uint32_t nq,
                                   int32_t* __restrict__ out_idx, float* __restrict__ out_d2) {
  const uint32_t qi = blockIdx.x;
  if (qi >= nq) return;
  const unsigned long long src = full_off[qi], dst = out_off[qi], cnt = out_off[qi + 1] - out_off[qi];
  for (unsigned long long t = threadIdx.x; t < cnt; t += blockDim.x) {
    const uint64_t k = keys[src + t];
    out_idx[dst + t] = int32_t(uint32_t(k));
    out_d2[dst + t] = __uint_as_float(uint32_t(k >> 32));
  }
}

struct Guard {
  std::vector<void*> p;
  ~Guard() {
    for (void* q : p)
      if (q) (void)hipFree(q);
  }
  template <class T>
  hipError_t alloc(T** ptr, size_t bytes) {
    hipError_t e = hipMalloc(ptr, bytes ? bytes : 16);
    if (e == hipSuccess) p.push_back(*ptr);
    return e;
  }
};

}  // namespace
}  // namespace pclhip

using namespace pclhip;

extern "C" pclhip_status pclhip_radius_search(pclhip_index* ix, const void* queries, size_t stride, uint64_t nq,
                                              double radius, uint32_t max_nn, uint64_t* out_offsets, int32_t* out_idx,
                                              float* out_d2, uint64_t capacity, uint64_t* out_total) {
  if (!ix || !out_offsets || !out_total) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = ix->ctx;
  *out_total = 0;
  PCLHIP_REQUIRE(ctx, stride >= 12 && stride % 4 == 0, "stride must be a multiple of 4 and >= 12 bytes");
  PCLHIP_REQUIRE(ctx, nq < 0x7FFFFFFFull, "too many queries");
  PCLHIP_REQUIRE(ctx, !is_device_pointer(out_offsets), "out_offsets must be host memory");
  out_offsets[0] = 0;
  if (nq == 0) return PCLHIP_OK;
  PCLHIP_REQUIRE(ctx, queries != nullptr, "null buffer");
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  Guard g;
  const void* dq = nullptr;
  void* owned = nullptr;
  pclhip_status st = to_device(ctx, queries, size_t(nq) * stride, &dq, &owned);
  if (st != PCLHIP_OK) return st;
  g.p.push_back(owned);
  float4* qs = nullptr;
  PCLHIP_CHECK_HIP(ctx, g.alloc(&qs, size_t(nq) * sizeof(float4)));
  uint32_t nf = 0;
  float lo[3], hi[3];
  st = spatial_order(ctx, dq, stride, nq, nullptr, 0, qs, uint32_t(nq), &nf, lo, hi, true, nullptr);
  if (st != PCLHIP_OK) return st;
  const float r2 = float(radius * radius);  // kdtree_flann.hpp:398
  const IndexView v = ix->view();
  const uint32_t n = uint32_t(nq);
  const uint32_t ngroups = (n + WAVE - 1) / WAVE;
  int grid = int((ngroups + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
  const int cap = ctx->num_cus * 4;
  if (grid > cap) grid = cap;
  if (grid < 1) grid = 1;
  uint32_t* counts = nullptr;
  unsigned long long *wide = nullptr, *full_off = nullptr, *out_off = nullptr;
  PCLHIP_CHECK_HIP(ctx, g.alloc(&counts, size_t(n) * 4));
  PCLHIP_CHECK_HIP(ctx, g.alloc(&wide, size_t(n + 1) * 8));
  PCLHIP_CHECK_HIP(ctx, g.alloc(&full_off, size_t(n + 1) * 8));
  PCLHIP_CHECK_HIP(ctx, g.alloc(&out_off, size_t(n + 1) * 8));
  hipLaunchKernelGGL(radius_kernel<false>, dim3(grid), dim3(BLOCK), 0, s, v, qs, n, r2, counts,
                     (const unsigned long long*)nullptr, (uint64_t*)nullptr);
  // exclusive scans of the full counts (segment starts) and of the clamped counts (output CSR)
  size_t tb = 0;
  PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(wide + n, 0, 8, s));
  PCLHIP_CHECK_HIP(ctx, rocprim::exclusive_scan(nullptr, tb, wide, full_off, 0ull, size_t(n + 1), rocprim::plus<unsigned long long>(), s));
  void* tmp = nullptr;
  PCLHIP_CHECK_HIP(ctx, g.alloc(&tmp, tb));
  hipLaunchKernelGGL(widen_counts_kernel, dim3((n + 255) / 256), dim3(256), 0, s, counts, n, wide);
  PCLHIP_CHECK_HIP(ctx, rocprim::exclusive_scan(tmp, tb, wide, full_off, 0ull, size_t(n + 1), rocprim::plus<unsigned long long>(), s));
  hipLaunchKernelGGL(clamp_counts_kernel, dim3((n + 255) / 256), dim3(256), 0, s, counts, n, max_nn, wide);
  PCLHIP_CHECK_HIP(ctx, rocprim::exclusive_scan(tmp, tb, wide, out_off, 0ull, size_t(n + 1), rocprim::plus<unsigned long long>(), s));
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(out_offsets, out_off, size_t(n + 1) * 8, hipMemcpyDeviceToHost, s));
  unsigned long long full_total = 0;
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(&full_total, full_off + n, 8, hipMemcpyDeviceToHost, s));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
  const uint64_t total = out_offsets[n];
  *out_total = total;
  if (total == 0) return PCLHIP_OK;
  if (!out_idx || !out_d2 || capacity < total) {
    set_error(ctx, "radius search: output capacity too small (out_total holds the required size)");
    return PCLHIP_ERR_OVERFLOW;
  }
  uint64_t *k0 = nullptr, *k1 = nullptr;
  PCLHIP_CHECK_HIP(ctx, g.alloc(&k0, size_t(full_total) * 8));
  PCLHIP_CHECK_HIP(ctx, g.alloc(&k1, size_t(full_total) * 8));
  hipLaunchKernelGGL(radius_kernel<true>, dim3(grid), dim3(BLOCK), 0, s, v, qs, n, r2, counts, full_off, k0);
  size_t sb = 0;
  PCLHIP_CHECK_HIP(ctx, rocprim::segmented_radix_sort_keys(nullptr, sb, k0, k1, size_t(full_total), n, full_off, full_off + 1, 0, 64, s));
  void* stmp = nullptr;
  PCLHIP_CHECK_HIP(ctx, g.alloc(&stmp, sb));
  PCLHIP_CHECK_HIP(ctx, rocprim::segmented_radix_sort_keys(stmp, sb, k0, k1, size_t(full_total), n, full_off, full_off + 1, 0, 64, s));
  int32_t* d_idx = out_idx;
  float* d_d2 = out_d2;
  const bool idx_dev = is_device_pointer(out_idx), d2_dev = is_device_pointer(out_d2);
  if (!idx_dev) PCLHIP_CHECK_HIP(ctx, g.alloc(&d_idx, size_t(total) * 4));
  if (!d2_dev) PCLHIP_CHECK_HIP(ctx, g.alloc(&d_d2, size_t(total) * 4));
  hipLaunchKernelGGL(radius_emit_kernel, dim3(n), dim3(64), 0, s, k1, full_off, out_off, n, d_idx, d_d2);
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  if (!idx_dev) PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(out_idx, d_idx, size_t(total) * 4, hipMemcpyDeviceToHost, s));
  if (!d2_dev) PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(out_d2, d_d2, size_t(total) * 4, hipMemcpyDeviceToHost, s));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
  return PCLHIP_OK;
}
