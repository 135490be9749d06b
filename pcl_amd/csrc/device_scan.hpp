// device_scan.hpp -- exclusive prefix sums of u32 arrays in three small launches (hand-written; used by the
// VoxelGrid run detection and radix sort, and by the compaction of clouds with non-finite points).
//   sc_partial_kernel  totals of every 4096-element block
//   sc_top_kernel      exclusive scan of the block totals (one workgroup), grand totals -> tot[0..1]
//   sc_apply_kernel    scan inside every block + its offset
// Two scans run at once: of a[i] ("sum") and of (a[i] != 0) ("runs"); either output may be null.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pclhip {
namespace {

constexpr int SC_BLOCK = 4096, SC_THREADS = 256, SC_PER = SC_BLOCK / SC_THREADS;

// Exclusive scans of a[i] ("sum") and of (a[i] != 0) ("runs") over m elements, in three launches:
// block totals -> scan of the totals (one workgroup) -> per-block scan.  tot[0..2] = total sum, total runs, max a[i].
__global__ __launch_bounds__(SC_THREADS) void sc_partial_kernel(const uint32_t* __restrict__ a, uint64_t m,
                                                                uint2* __restrict__ partial, uint32_t* __restrict__ tot) {
  const uint64_t base = uint64_t(blockIdx.x) * SC_BLOCK + uint64_t(threadIdx.x) * SC_PER;
  uint32_t s = 0, r = 0, mx = 0;
#pragma unroll
  for (int e = 0; e < SC_PER; ++e) {
    const uint64_t i = base + e;
    const uint32_t v = i < m ? a[i] : 0u;
    s += v;
    r += v != 0u ? 1u : 0u;
    mx = v > mx ? v : mx;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    r += __shfl_xor(r, o);
    const uint32_t t = __shfl_xor(mx, o);
    mx = t > mx ? t : mx;
  }
  __shared__ uint32_t ws[SC_THREADS / 64][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { ws[wave][0] = s; ws[wave][1] = r; ws[wave][2] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t S = 0, R = 0, M = 0;
    for (int w = 0; w < SC_THREADS / 64; ++w) { S += ws[w][0]; R += ws[w][1]; M = ws[w][2] > M ? ws[w][2] : M; }
    partial[blockIdx.x] = make_uint2(S, R);
    if (M) atomicMax(tot + 2, M);
  }
}

__global__ __launch_bounds__(1024) void sc_top_kernel(uint2* __restrict__ partial, uint32_t nb, uint32_t* __restrict__ tot) {
  __shared__ uint2 sh[1024];
  uint2 carry = make_uint2(0u, 0u);
  for (uint32_t b0 = 0; b0 < nb; b0 += 1024) {
    const uint32_t i = b0 + threadIdx.x;
    const uint2 own = i < nb ? partial[i] : make_uint2(0u, 0u);
    sh[threadIdx.x] = own;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      uint2 v = make_uint2(0u, 0u);
      if (threadIdx.x >= uint32_t(o)) v = sh[threadIdx.x - o];
      __syncthreads();
      sh[threadIdx.x].x += v.x;
      sh[threadIdx.x].y += v.y;
      __syncthreads();
    }
    const uint2 incl = sh[threadIdx.x];
    if (i < nb) partial[i] = make_uint2(carry.x + incl.x - own.x, carry.y + incl.y - own.y);
    const uint2 last = sh[1023];
    __syncthreads();
    carry.x += last.x;
    carry.y += last.y;
  }
  if (threadIdx.x == 0) {
    tot[0] = carry.x;
    tot[1] = carry.y;
  }
}

// out_sum[i], out_run[i] (either may be null); run_start (optional): run_start[run of i] = sum before i for every
// non-zero a[i], plus the end sentinel run_start[total runs] = total sum
__global__ __launch_bounds__(SC_THREADS) void sc_apply_kernel(const uint32_t* __restrict__ a, uint64_t m,
                                                              const uint2* __restrict__ partial, uint32_t* __restrict__ out_sum,
                                                              uint32_t* __restrict__ out_run, uint32_t* __restrict__ run_start) {
  const uint64_t base = uint64_t(blockIdx.x) * SC_BLOCK + uint64_t(threadIdx.x) * SC_PER;
  uint32_t v[SC_PER];
  uint32_t s = 0, r = 0;
#pragma unroll
  for (int e = 0; e < SC_PER; ++e) {
    const uint64_t i = base + e;
    v[e] = i < m ? a[i] : 0u;
    s += v[e];
    r += v[e] != 0u ? 1u : 0u;
  }
  // exclusive scan of (s, r) over the workgroup's threads
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t is = s, ir = r;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t ts = __shfl_up(is, o), tr = __shfl_up(ir, o);
    if (lane >= o) { is += ts; ir += tr; }
  }
  __shared__ uint32_t ws[SC_THREADS / 64][2];
  if (lane == 63) { ws[wave][0] = is; ws[wave][1] = ir; }
  __syncthreads();
  uint32_t bs = 0, br = 0;
  for (int w = 0; w < wave; ++w) { bs += ws[w][0]; br += ws[w][1]; }
  const uint2 p = partial[blockIdx.x];
  uint32_t run_s = p.x + bs + is - s, run_r = p.y + br + ir - r;
#pragma unroll
  for (int e = 0; e < SC_PER; ++e) {
    const uint64_t i = base + e;
    if (i < m) {
      if (out_sum) out_sum[i] = run_s;
      if (out_run) out_run[i] = run_r;
      if (run_start && v[e] != 0u) run_start[run_r] = run_s;
      run_s += v[e];
      run_r += v[e] != 0u ? 1u : 0u;
      if (run_start && i == m - 1) run_start[run_r] = run_s;
    }
  }
}

// exclusive scan of a[0..m) on stream s; partial: ceil(m / SC_BLOCK) uint2 of scratch; tot: 4 u32 (tot[0] = sum,
// tot[1] = number of non-zero elements after the launches)
inline void launch_scan_u32(hipStream_t s, const uint32_t* a, uint64_t m, uint2* partial, uint32_t* tot, uint32_t* out_sum,
                            uint32_t* out_run = nullptr, uint32_t* run_start = nullptr) {
  const uint32_t blocks = uint32_t((m + SC_BLOCK - 1) / SC_BLOCK);
  hipLaunchKernelGGL(sc_partial_kernel, dim3(blocks), dim3(SC_THREADS), 0, s, a, m, partial, tot);
  hipLaunchKernelGGL(sc_top_kernel, dim3(1), dim3(1024), 0, s, partial, blocks, tot);
  hipLaunchKernelGGL(sc_apply_kernel, dim3(blocks), dim3(SC_THREADS), 0, s, a, m, partial, out_sum, out_run, run_start);
}

}  // namespace
}  // namespace pclhip
