// shard_dev.hip -- the partition and the halo selection of shard.cpp for clouds that ARE in device memory: the same
// kd cells (same cut values, bit for bit) and the same ascending index lists, without the cloud ever visiting the
// host.  (PCL has no multi-GPU path.  Host clouds keep the host code of shard.cpp -- it runs without a
// GPU, which is what the multi-process CPU tests exercise; tests compare the two.)
//
// bisect() of shard.cpp needs, per cell: the bounding box of its points (widest axis), ONE order statistic of one
// coordinate (the cut; its value does not depend on how ties are ordered) and the number of points strictly below it.
// Here a cell is a byte label per point; per cell: one bounding-box pass, three histogram passes of a radix selection
// over the order-preserving key of the coordinate (10 + 11 + 11 bits; the remaining rank inside the last bin tells how
// many points are strictly below the cut), one relabel pass.  Every pass streams the records once (16 B per point at
// PointXYZ stride): 35 passes for 8 slabs.  The selection of a region's points is flag -> scan -> ordered scatter.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "pclhip_internal.hpp"
#include "device_scan.hpp"

namespace pclhip {
namespace {

constexpr int TB = 256;
constexpr int BINS = 2048;
constexpr uint8_t NO_CELL = 0xFF;

// order-preserving key of a float; -0.0 counts as +0.0 (the host compares floats)
__device__ __forceinline__ uint32_t fkey(float v) {
  const uint32_t b = __float_as_uint(v + 0.0f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
inline float key_to_float(uint32_t k) {
  const uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  float f;
  std::memcpy(&f, &b, 4);
  return f;
}

__device__ __forceinline__ const float* rec(const char* base, size_t stride, uint64_t i) {
  return reinterpret_cast<const float*>(base + size_t(i) * stride);
}

struct SelState {
  uint32_t key;    // key bits decided so far
  uint32_t rank;   // remaining rank inside the chosen bin
  uint32_t below;  // points of the cell strictly below the chosen bin(s)
  uint32_t pad;
};

// cell[i] = 0 for finite points, NO_CELL otherwise; *count = finite points
__global__ __launch_bounds__(TB) void shard_init_kernel(const char* __restrict__ pts, size_t stride, uint64_t n,
                                                        uint8_t* __restrict__ cell, unsigned int* __restrict__ count) {
  unsigned int c = 0;
  for (uint64_t i = uint64_t(blockIdx.x) * TB + threadIdx.x; i < n; i += uint64_t(gridDim.x) * TB) {
    const float* p = rec(pts, stride, i);
    const bool ok = isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]);
    cell[i] = ok ? 0 : NO_CELL;
    c += ok ? 1u : 0u;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

// bounding box of cell `id` as keys: box[0..2] = min, box[3..5] = max
__global__ __launch_bounds__(TB) void shard_bbox_kernel(const char* __restrict__ pts, size_t stride, uint64_t n,
                                                        const uint8_t* __restrict__ cell, uint8_t id,
                                                        uint32_t* __restrict__ box) {
  uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
  for (uint64_t i = uint64_t(blockIdx.x) * TB + threadIdx.x; i < n; i += uint64_t(gridDim.x) * TB) {
    if (cell[i] != id) continue;
    const float* p = rec(pts, stride, i);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const uint32_t k = fkey(p[d]);
      mn[d] = k < mn[d] ? k : mn[d];
      mx[d] = k > mx[d] ? k : mx[d];
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t a = __shfl_xor(mn[d], o), b = __shfl_xor(mx[d], o);
      mn[d] = a < mn[d] ? a : mn[d];
      mx[d] = b > mx[d] ? b : mx[d];
    }
    if ((threadIdx.x & 63) == 0) {
      if (mn[d] != 0xFFFFFFFFu) atomicMin(box + d, mn[d]);
      if (mx[d] != 0u) atomicMax(box + 3 + d, mx[d]);
    }
  }
}

template <int PASS>
__global__ __launch_bounds__(TB) void shard_hist_kernel(const char* __restrict__ pts, size_t stride, uint64_t n,
                                                        const uint8_t* __restrict__ cell, uint8_t id, int axis,
                                                        const SelState* __restrict__ st, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[BINS];
  for (int i = threadIdx.x; i < BINS; i += TB) h[i] = 0u;
  __syncthreads();
  const uint32_t key = PASS > 0 ? st->key : 0u;
  for (uint64_t i = uint64_t(blockIdx.x) * TB + threadIdx.x; i < n; i += uint64_t(gridDim.x) * TB) {
    if (cell[i] != id) continue;
    const uint32_t k = fkey(rec(pts, stride, i)[axis]);
    if constexpr (PASS == 0) {
      atomicAdd(&h[k >> 22], 1u);
    } else if constexpr (PASS == 1) {
      if ((k >> 22) == (key >> 22)) atomicAdd(&h[(k >> 11) & 2047u], 1u);
    } else {
      if ((k >> 11) == (key >> 11)) atomicAdd(&h[k & 2047u], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < BINS; i += TB)
    if (h[i]) atomicAdd(hist + i, h[i]);
}

template <int PASS>
__global__ __launch_bounds__(256) void shard_pick_kernel(SelState* __restrict__ st, const uint32_t* __restrict__ hist) {
  constexpr int PER = BINS / 256;
  uint32_t c[PER], sum = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    c[j] = hist[threadIdx.x * PER + j];
    sum += c[j];
  }
  __shared__ uint32_t scan[256];
  scan[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const uint32_t v = threadIdx.x >= uint32_t(o) ? scan[threadIdx.x - o] : 0u;
    __syncthreads();
    scan[threadIdx.x] += v;
    __syncthreads();
  }
  const uint32_t rank = st->rank, key = st->key, below = st->below;
  const uint32_t incl = scan[threadIdx.x], excl = incl - sum;
  __syncthreads();
  if (rank >= excl && rank < incl) {
    uint32_t cum = excl;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      if (rank >= cum && rank < cum + c[j]) {
        constexpr int SHIFT = PASS == 0 ? 22 : PASS == 1 ? 11 : 0;
        st->key = key | (uint32_t(threadIdx.x * PER + j) << SHIFT);
        st->rank = rank - cum;
        st->below = below + cum;
      }
      cum += c[j];
    }
  }
}

// points of cell `id` at or above the cut move to cell `to`
__global__ __launch_bounds__(TB) void shard_relabel_kernel(const char* __restrict__ pts, size_t stride, uint64_t n,
                                                           uint8_t* __restrict__ cell, uint8_t id, int axis,
                                                           const SelState* __restrict__ st, uint8_t to) {
  const uint32_t cut = st->key;
  for (uint64_t i = uint64_t(blockIdx.x) * TB + threadIdx.x; i < n; i += uint64_t(gridDim.x) * TB) {
    if (cell[i] != id) continue;
    if (fkey(rec(pts, stride, i)[axis]) >= cut) cell[i] = to;
  }
}

// flag[i] = the point lies inside the (closed) box; non-finite coordinates fail every comparison
__global__ __launch_bounds__(TB) void shard_flag_kernel(const char* __restrict__ pts, size_t stride, uint64_t n, float lx,
                                                        float ly, float lz, float hx, float hy, float hz,
                                                        uint32_t* __restrict__ flag) {
  const uint64_t i = uint64_t(blockIdx.x) * TB + threadIdx.x;
  if (i >= n) return;
  const float* p = rec(pts, stride, i);
  flag[i] = (p[0] >= lx && p[0] <= hx && p[1] >= ly && p[1] <= hy && p[2] >= lz && p[2] <= hz) ? 1u : 0u;
}
__global__ __launch_bounds__(TB) void shard_emit_kernel(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos,
                                                        uint64_t n, int32_t* __restrict__ out) {
  const uint64_t i = uint64_t(blockIdx.x) * TB + threadIdx.x;
  if (i < n && flag[i]) out[pos[i]] = int32_t(i);
}

struct Bufs {
  std::vector<void*> p;
  ~Bufs() {
    for (void* q : p)
      if (q) (void)hipFree(q);
  }
  template <class T>
  hipError_t alloc(T** ptr, size_t bytes) {
    void* q = nullptr;
    const hipError_t e = hipMalloc(&q, bytes ? bytes : 16);
    if (e == hipSuccess) p.push_back(q);
    *ptr = static_cast<T*>(q);
    return e;
  }
};

#define SH_CHECK(expr)                                              \
  do {                                                              \
    const hipError_t e_ = (expr);                                   \
    if (e_ != hipSuccess) {                                         \
      set_error(nullptr, hipGetErrorString(e_));                    \
      return PCLHIP_ERR_HIP;                                        \
    }                                                               \
  } while (0)

int grid_for(uint64_t n) {  // grid-stride passes: eight workgroups per CU of an MI355X
  uint64_t g = (n + TB - 1) / TB;
  if (g > 2048) g = 2048;
  return g < 1 ? 1 : int(g);
}

struct Cell {
  uint8_t id;
  int parts, out;  // the cell becomes slabs [out, out + parts)
  uint32_t count;
  float lo[3], hi[3];
};

}  // namespace

// pclhip_partition_slabs for a cloud in device memory (same regions as shard.cpp's bisect(), bit for bit)
pclhip_status partition_slabs_device(const void* points, size_t stride, uint64_t n, int n_slabs, float* regions) {
  const float inf = std::numeric_limits<float>::infinity();
  // These setup calls take no context: they run on the null stream.  A cloud that another (non-blocking) stream is still
  // writing is not ordered before them by the stream semantics, so everything queued on the device is waited for first.
  if (hipDeviceSynchronize() != hipSuccess) {
    set_error(nullptr, "device synchronisation failed before the partition");
    return PCLHIP_ERR_HIP;
  }
  if (n_slabs > 254) {
    set_error(nullptr, "at most 254 slabs");
    return PCLHIP_ERR_INVALID;
  }
  const char* pts = static_cast<const char*>(points);
  Bufs b;
  uint8_t* cell = nullptr;
  uint32_t *hist = nullptr, *box = nullptr;
  SelState* st = nullptr;
  unsigned int* cnt = nullptr;
  SH_CHECK(b.alloc(&cell, size_t(n)));
  SH_CHECK(b.alloc(&hist, 3 * BINS * sizeof(uint32_t)));
  SH_CHECK(b.alloc(&box, 6 * sizeof(uint32_t)));
  SH_CHECK(b.alloc(&st, sizeof(SelState)));
  SH_CHECK(b.alloc(&cnt, sizeof(unsigned int)));
  hipStream_t s = nullptr;  // the partition runs once per target, on the legacy stream
  const int grid = grid_for(n);
  SH_CHECK(hipMemsetAsync(cnt, 0, sizeof(unsigned int), s));
  hipLaunchKernelGGL(shard_init_kernel, dim3(grid), dim3(TB), 0, s, pts, stride, n, cell, cnt);
  unsigned int finite = 0;
  SH_CHECK(hipMemcpy(&finite, cnt, sizeof(finite), hipMemcpyDeviceToHost));
  std::vector<Cell> todo;
  Cell root = {0, n_slabs, 0, finite, {-inf, -inf, -inf}, {inf, inf, inf}};
  todo.push_back(root);
  auto emit = [&](int g, const float* lo, const float* hi) {
    std::memcpy(regions + 6 * g, lo, 3 * sizeof(float));
    std::memcpy(regions + 6 * g + 3, hi, 3 * sizeof(float));
  };
  while (!todo.empty()) {
    const Cell c = todo.back();
    todo.pop_back();
    if (c.parts == 1) {
      emit(c.out, c.lo, c.hi);
      continue;
    }
    if (c.count == 0) {  // no points to cut: the first part keeps the whole cell, the others an empty one (shard.cpp)
      emit(c.out, c.lo, c.hi);
      for (int p = 1; p < c.parts; ++p) {
        float lo[3] = {c.hi[0], c.lo[1], c.lo[2]};
        emit(c.out + p, lo, c.hi);
      }
      continue;
    }
    // widest axis of the points themselves
    const uint32_t box0[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    SH_CHECK(hipMemcpyAsync(box, box0, sizeof(box0), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(shard_bbox_kernel, dim3(grid), dim3(TB), 0, s, pts, stride, n, cell, c.id, box);
    uint32_t hb[6];
    SH_CHECK(hipMemcpy(hb, box, sizeof(hb), hipMemcpyDeviceToHost));
    float mn[3], mx[3];
    for (int d = 0; d < 3; ++d) {
      mn[d] = key_to_float(hb[d]);
      mx[d] = key_to_float(hb[3 + d]);
    }
    int axis = 0;
    float best = mx[0] - mn[0];
    if (mx[1] - mn[1] > best) { best = mx[1] - mn[1]; axis = 1; }
    if (mx[2] - mn[2] > best) axis = 2;
    const int left_parts = c.parts / 2, right_parts = c.parts - left_parts;
    uint64_t k = uint64_t(c.count) * uint64_t(left_parts) / uint64_t(c.parts);
    if (k >= c.count) k = c.count - 1;
    const SelState st0 = {0u, uint32_t(k), 0u, 0u};
    SH_CHECK(hipMemcpyAsync(st, &st0, sizeof(st0), hipMemcpyHostToDevice, s));
    SH_CHECK(hipMemsetAsync(hist, 0, 3 * BINS * sizeof(uint32_t), s));
    hipLaunchKernelGGL(shard_hist_kernel<0>, dim3(grid), dim3(TB), 0, s, pts, stride, n, cell, c.id, axis, st, hist);
    hipLaunchKernelGGL(shard_pick_kernel<0>, dim3(1), dim3(256), 0, s, st, hist);
    hipLaunchKernelGGL(shard_hist_kernel<1>, dim3(grid), dim3(TB), 0, s, pts, stride, n, cell, c.id, axis, st, hist + BINS);
    hipLaunchKernelGGL(shard_pick_kernel<1>, dim3(1), dim3(256), 0, s, st, hist + BINS);
    hipLaunchKernelGGL(shard_hist_kernel<2>, dim3(grid), dim3(TB), 0, s, pts, stride, n, cell, c.id, axis, st,
                       hist + 2 * BINS);
    hipLaunchKernelGGL(shard_pick_kernel<2>, dim3(1), dim3(256), 0, s, st, hist + 2 * BINS);
    const uint8_t right_id = uint8_t(c.out + left_parts);  // a cell's label = its first slab
    hipLaunchKernelGGL(shard_relabel_kernel, dim3(grid), dim3(TB), 0, s, pts, stride, n, cell, c.id, axis, st, right_id);
    SelState got;
    SH_CHECK(hipMemcpy(&got, st, sizeof(got), hipMemcpyDeviceToHost));
    const float cut = key_to_float(got.key);
    // left = strictly below the cut (ties go right, like the kernel's x >= lo && x < hi ownership test)
    Cell L = c, R = c;
    L.hi[axis] = cut;
    L.parts = left_parts;
    L.count = got.below;
    R.lo[axis] = cut;
    R.parts = right_parts;
    R.out = c.out + left_parts;
    R.id = right_id;
    R.count = c.count - got.below;
    todo.push_back(R);
    todo.push_back(L);
  }
  SH_CHECK(hipGetLastError());
  return PCLHIP_OK;
}

// pclhip_select_region for a cloud in device memory: lo / hi are the dilated, outward-rounded bounds of shard.cpp
pclhip_status select_region_device(const void* points, size_t stride, uint64_t n, const float lo[3], const float hi[3],
                                   int32_t* out_indices, uint64_t capacity, uint64_t* out_count) {
  if (hipDeviceSynchronize() != hipSuccess) {  // see partition_slabs_device
    set_error(nullptr, "device synchronisation failed before the selection");
    return PCLHIP_ERR_HIP;
  }
  *out_count = 0;
  if (n == 0) return PCLHIP_OK;
  const char* pts = static_cast<const char*>(points);
  Bufs b;
  uint32_t *flag = nullptr, *pos = nullptr, *tot = nullptr;
  uint2* partial = nullptr;
  SH_CHECK(b.alloc(&flag, size_t(n) * 4));
  SH_CHECK(b.alloc(&pos, size_t(n) * 4));
  SH_CHECK(b.alloc(&partial, size_t((n + SC_BLOCK - 1) / SC_BLOCK) * sizeof(uint2)));
  SH_CHECK(b.alloc(&tot, 4 * sizeof(uint32_t)));
  hipStream_t s = nullptr;
  const dim3 grid(uint32_t((n + TB - 1) / TB)), block(TB);
  SH_CHECK(hipMemsetAsync(tot, 0, 4 * sizeof(uint32_t), s));
  hipLaunchKernelGGL(shard_flag_kernel, grid, block, 0, s, pts, stride, n, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], flag);
  launch_scan_u32(s, flag, n, partial, tot, pos);
  uint32_t total = 0;
  SH_CHECK(hipMemcpy(&total, tot, sizeof(total), hipMemcpyDeviceToHost));
  *out_count = total;
  if (total > capacity || (total > 0 && !out_indices)) {
    set_error(nullptr, "index buffer too small for the selected points");
    return PCLHIP_ERR_OVERFLOW;
  }
  if (total == 0) return PCLHIP_OK;
  int32_t* d_out = out_indices;
  const bool dev = is_device_pointer(out_indices);
  if (!dev) SH_CHECK(b.alloc(&d_out, size_t(total) * 4));
  hipLaunchKernelGGL(shard_emit_kernel, grid, block, 0, s, flag, pos, n, d_out);
  SH_CHECK(hipGetLastError());
  if (!dev) SH_CHECK(hipMemcpy(out_indices, d_out, size_t(total) * 4, hipMemcpyDeviceToHost));
  else SH_CHECK(hipStreamSynchronize(s));
  return PCLHIP_OK;
}

}  // namespace pclhip
