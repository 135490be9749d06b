// voxelgrid.hip -- pcl::VoxelGrid<pcl::PointXYZ>::applyFilter on the GPU
// (filters/include/pcl/filters/impl/voxel_grid.hpp:597-814): bounding box -> int32 voxel id per
// point (same float expressions, :713-718) -> stable radix sort of (voxel id, point index) -> run
// boundaries -> per-voxel centroid (float sum / count, common/include/pcl/common/impl/
// accumulators.hpp:68-85) in ascending voxel id order.  Within a voxel the points are summed in
// ascending input index order (the stable sort fixes what the reference's spreadsort leaves
// unspecified), so centroids are bit-identical to the CPU oracle.
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include <cfloat>
#include <cmath>

#include "pclhip_internal.hpp"

namespace pclhip {
namespace {

__device__ __forceinline__ const float* rec(const void* base, size_t stride, uint64_t i) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + i * stride);
}

// getMinMax3D with the optional field filter (voxel_grid.hpp:513-590; limits cast to float, :615)
__global__ __launch_bounds__(256) void vg_minmax_kernel(const void* pts, size_t stride, uint64_t n, int has_limits,
                                                        float fmin_, float fmax_, float* partial) {
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (uint64_t i = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x; i < n; i += uint64_t(gridDim.x) * blockDim.x) {
    const float* p = rec(pts, stride, i);
    const float x = p[0], y = p[1], z = p[2];
    if (has_limits && ((z > fmax_) || (z < fmin_))) continue;
    if (!(isfinite(x) && isfinite(y) && isfinite(z))) continue;
    lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
    hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
  }
  __shared__ float s[4][6];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float a = lo[d], b = hi[d];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      a = fminf(a, __shfl_xor(a, o));
      b = fmaxf(b, __shfl_xor(b, o));
    }
    if (lane == 0) {
      s[wave][d] = a;
      s[wave][3 + d] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = s[0][threadIdx.x];
    for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fminf(v, s[w][threadIdx.x]) : fmaxf(v, s[w][threadIdx.x]);
    partial[blockIdx.x * 6 + threadIdx.x] = v;
  }
}

struct VgGrid {
  float inv[3];
  int min_b[3];
  int mul[3];
};

__global__ __launch_bounds__(256) void vg_key_kernel(const void* pts, size_t stride, uint64_t n, VgGrid g, int has_limits,
                                                     double lim_min, double lim_max, uint32_t* keys, uint32_t* vals,
                                                     unsigned int* n_valid) {
  const uint64_t i = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
  bool ok = false;
  if (i < n) {
    const float* p = rec(pts, stride, i);
    const float x = p[0], y = p[1], z = p[2];
    ok = isfinite(x) && isfinite(y) && isfinite(z);
    if (ok && has_limits) ok = !((double(z) > lim_max) || (double(z) < lim_min));  // :684-695
    uint32_t key = 0xFFFFFFFFu;
    if (ok) {  // :713-718
      const int i0 = int(floorf(__fmul_rn(x, g.inv[0])) - float(g.min_b[0]));
      const int i1 = int(floorf(__fmul_rn(y, g.inv[1])) - float(g.min_b[1]));
      const int i2 = int(floorf(__fmul_rn(z, g.inv[2])) - float(g.min_b[2]));
      key = uint32_t(i0 * g.mul[0] + i1 * g.mul[1] + i2 * g.mul[2]);
    }
    keys[i] = key;
    vals[i] = uint32_t(i);
  }
  const unsigned long long b = __builtin_amdgcn_ballot_w64(ok);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_valid, (unsigned int)__builtin_popcountll(b));
}

__global__ void vg_head_kernel(const uint32_t* keys, uint32_t nv, uint32_t* head) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < nv) head[j] = (j == 0 || keys[j] != keys[j - 1]) ? 1u : 0u;
}

// run_start[r] = first sorted position of run r (r = inclusive_scan(head) - 1)
__global__ void vg_runstart_kernel(const uint32_t* head, const uint32_t* scan, uint32_t nv, uint32_t* run_start) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < nv && head[j]) run_start[scan[j] - 1] = j;
  if (j == nv - 1) run_start[scan[j]] = nv;  // end sentinel
}

__global__ void vg_keep_kernel(const uint32_t* run_start, uint32_t nruns, uint32_t min_pts, uint32_t* keep) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < nruns) keep[r] = ((run_start[r + 1] - run_start[r]) >= min_pts) ? 1u : 0u;
}

// one thread per voxel: sequential float accumulation in sorted (= ascending input index) order
__global__ __launch_bounds__(256) void vg_centroid_kernel(const void* pts, size_t stride, const uint32_t* vals,
                                                          const uint32_t* run_start, const uint32_t* keep,
                                                          const uint32_t* keep_scan, uint32_t nruns, float4* out) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nruns || !keep[r]) return;
  const uint32_t b = run_start[r], e = run_start[r + 1];
  float sx = 0.0f, sy = 0.0f, sz = 0.0f;
  for (uint32_t j = b; j < e; ++j) {
    const float* p = rec(pts, stride, vals[j]);
    sx = __fadd_rn(sx, p[0]);
    sy = __fadd_rn(sy, p[1]);
    sz = __fadd_rn(sz, p[2]);
  }
  const float cnt = float(e - b);
  out[keep_scan[r] - 1] = make_float4(__fdiv_rn(sx, cnt), __fdiv_rn(sy, cnt), __fdiv_rn(sz, cnt), 1.0f);
}

}  // namespace
}  // namespace pclhip

using namespace pclhip;

extern "C" pclhip_status pclhip_voxelgrid(pclhip_ctx* ctx, const void* points, size_t stride, uint64_t n,
                                          const float leaf[3], uint32_t min_points_per_voxel, int has_z_limits,
                                          double z_min, double z_max, void* out_xyzw, uint64_t* out_n) {
  if (!ctx || !leaf || !out_n) return PCLHIP_ERR_INVALID;
  *out_n = 0;
  PCLHIP_REQUIRE(ctx, stride >= 12 && stride % 4 == 0, "stride must be a multiple of 4 and >= 12 bytes");
  PCLHIP_REQUIRE(ctx, n < 0x7FFFFFFFull, "cloud too large for int32 indices");
  PCLHIP_REQUIRE(ctx, leaf[0] > 0 && leaf[1] > 0 && leaf[2] > 0, "leaf size must be positive");
  if (n == 0) return PCLHIP_OK;
  PCLHIP_REQUIRE(ctx, points && out_xyzw, "null buffer");
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  struct Guard {
    std::vector<void*> p;
    ~Guard() {
      for (void* q : p)
        if (q) (void)hipFree(q);
    }
  } guard;
  const void* dp = nullptr;
  void* owned = nullptr;
  pclhip_status st = to_device(ctx, points, size_t(n) * stride, &dp, &owned);
  if (st != PCLHIP_OK) return st;
  guard.p.push_back(owned);

  // --- bounding box (getMinMax3D) ---
  int nb = int((n + 2047) / 2048);
  if (nb > 1024) nb = 1024;
  float* d_partial = nullptr;
  PCLHIP_CHECK_HIP(ctx, hipMalloc(&d_partial, size_t(nb) * 6 * sizeof(float)));
  guard.p.push_back(d_partial);
  hipLaunchKernelGGL(vg_minmax_kernel, dim3(nb), dim3(256), 0, s, dp, stride, n, has_z_limits, float(z_min), float(z_max),
                     d_partial);
  std::vector<float> hp(size_t(nb) * 6);
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(hp.data(), d_partial, hp.size() * sizeof(float), hipMemcpyDeviceToHost, s));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int b = 0; b < nb; ++b)
    for (int d = 0; d < 3; ++d) {
      mn[d] = std::fmin(mn[d], hp[size_t(b) * 6 + d]);
      mx[d] = std::fmax(mx[d], hp[size_t(b) * 6 + 3 + d]);
    }
  VgGrid g;
  for (int d = 0; d < 3; ++d) g.inv[d] = 1.0f / leaf[d];  // voxel_grid.h:279-282
  // :620-629 overflow check (float product, then int64)
  volatile float ex = (mx[0] - mn[0]) * g.inv[0], ey = (mx[1] - mn[1]) * g.inv[1], ez = (mx[2] - mn[2]) * g.inv[2];
  const int64_t dx = int64_t(ex) + 1, dy = int64_t(ey) + 1, dz = int64_t(ez) + 1;
  if (dx * dy * dz > int64_t(INT32_MAX)) {
    set_error(ctx, "VoxelGrid: leaf size too small for the input dataset, integer indices would overflow");
    return PCLHIP_ERR_OVERFLOW;
  }
  int div_b[3];
  for (int d = 0; d < 3; ++d) {  // :632-640
    volatile float a = mn[d] * g.inv[d], b = mx[d] * g.inv[d];
    g.min_b[d] = int(std::floor(a));
    const int max_b = int(std::floor(b));
    div_b[d] = max_b - g.min_b[d] + 1;
  }
  g.mul[0] = 1;
  g.mul[1] = div_b[0];
  g.mul[2] = div_b[0] * div_b[1];

  // --- keys + stable sort ---
  uint32_t *k0, *k1, *v0, *v1, *head, *scan, *run_start, *keep, *keep_scan;
  unsigned int* d_cnt;
  PCLHIP_CHECK_HIP(ctx, hipMalloc(&k0, n * sizeof(uint32_t))); guard.p.push_back(k0);
  PCLHIP_CHECK_HIP(ctx, hipMalloc(&k1, n * sizeof(uint32_t))); guard.p.push_back(k1);
  PCLHIP_CHECK_HIP(ctx, hipMalloc(&v0, n * sizeof(uint32_t))); guard.p.push_back(v0);
  PCLHIP_CHECK_HIP(ctx, hipMalloc(&v1, n * sizeof(uint32_t))); guard.p.push_back(v1);
  PCLHIP_CHECK_HIP(ctx, hipMalloc(&d_cnt, sizeof(unsigned int))); guard.p.push_back(d_cnt);
  PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(d_cnt, 0, sizeof(unsigned int), s));
  hipLaunchKernelGGL(vg_key_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, dp, stride, n, g, has_z_limits, z_min,
                     z_max, k0, v0, d_cnt);
  size_t temp_bytes = 0;
  PCLHIP_CHECK_HIP(ctx, rocprim::radix_sort_pairs(nullptr, temp_bytes, k0, k1, v0, v1, size_t(n), 0, 32, s));
  void* tmp = nullptr;
  PCLHIP_CHECK_HIP(ctx, hipMalloc(&tmp, temp_bytes > 0 ? temp_bytes : 16)); guard.p.push_back(tmp);
  PCLHIP_CHECK_HIP(ctx, rocprim::radix_sort_pairs(tmp, temp_bytes, k0, k1, v0, v1, size_t(n), 0, 32, s));
  unsigned int nv = 0;
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(&nv, d_cnt, sizeof nv, hipMemcpyDeviceToHost, s));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
  if (nv == 0) return PCLHIP_OK;

  // --- runs ---
  PCLHIP_CHECK_HIP(ctx, hipMalloc(&head, size_t(nv) * sizeof(uint32_t))); guard.p.push_back(head);
  PCLHIP_CHECK_HIP(ctx, hipMalloc(&scan, size_t(nv) * sizeof(uint32_t))); guard.p.push_back(scan);
  hipLaunchKernelGGL(vg_head_kernel, dim3((nv + 255) / 256), dim3(256), 0, s, k1, nv, head);
  size_t scan_bytes = 0;
  PCLHIP_CHECK_HIP(ctx, rocprim::inclusive_scan(nullptr, scan_bytes, head, scan, size_t(nv), rocprim::plus<uint32_t>(), s));
  void* stmp = nullptr;
  PCLHIP_CHECK_HIP(ctx, hipMalloc(&stmp, scan_bytes > 0 ? scan_bytes : 16)); guard.p.push_back(stmp);
  PCLHIP_CHECK_HIP(ctx, rocprim::inclusive_scan(stmp, scan_bytes, head, scan, size_t(nv), rocprim::plus<uint32_t>(), s));
  uint32_t nruns = 0;
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(&nruns, scan + (nv - 1), sizeof nruns, hipMemcpyDeviceToHost, s));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
  PCLHIP_CHECK_HIP(ctx, hipMalloc(&run_start, size_t(nruns + 1) * sizeof(uint32_t))); guard.p.push_back(run_start);
  PCLHIP_CHECK_HIP(ctx, hipMalloc(&keep, size_t(nruns) * sizeof(uint32_t))); guard.p.push_back(keep);
  PCLHIP_CHECK_HIP(ctx, hipMalloc(&keep_scan, size_t(nruns) * sizeof(uint32_t))); guard.p.push_back(keep_scan);
  hipLaunchKernelGGL(vg_runstart_kernel, dim3((nv + 255) / 256), dim3(256), 0, s, head, scan, nv, run_start);
  hipLaunchKernelGGL(vg_keep_kernel, dim3((nruns + 255) / 256), dim3(256), 0, s, run_start, nruns, min_points_per_voxel,
                     keep);
  size_t scan2_bytes = 0;
  PCLHIP_CHECK_HIP(ctx,
                   rocprim::inclusive_scan(nullptr, scan2_bytes, keep, keep_scan, size_t(nruns), rocprim::plus<uint32_t>(), s));
  void* stmp2 = nullptr;
  PCLHIP_CHECK_HIP(ctx, hipMalloc(&stmp2, scan2_bytes > 0 ? scan2_bytes : 16)); guard.p.push_back(stmp2);
  PCLHIP_CHECK_HIP(ctx,
                   rocprim::inclusive_scan(stmp2, scan2_bytes, keep, keep_scan, size_t(nruns), rocprim::plus<uint32_t>(), s));
  uint32_t total = 0;
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(&total, keep_scan + (nruns - 1), sizeof total, hipMemcpyDeviceToHost, s));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));

  // --- centroids ---
  float4* d_out = static_cast<float4*>(out_xyzw);
  const bool out_dev = is_device_pointer(out_xyzw);
  if (!out_dev) {
    PCLHIP_CHECK_HIP(ctx, hipMalloc(&d_out, size_t(total > 0 ? total : 1) * sizeof(float4)));
    guard.p.push_back(d_out);
  }
  if (total > 0) {
    hipLaunchKernelGGL(vg_centroid_kernel, dim3((nruns + 255) / 256), dim3(256), 0, s, dp, stride, v1, run_start, keep,
                       keep_scan, nruns, d_out);
    PCLHIP_CHECK_HIP(ctx, hipGetLastError());
    if (!out_dev)
      PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(out_xyzw, d_out, size_t(total) * sizeof(float4), hipMemcpyDeviceToHost, s));
  }
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
  *out_n = total;
  return PCLHIP_OK;
}
