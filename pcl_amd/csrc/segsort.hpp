// segsort.hpp -- the two primitives of radiusSearch's CSR lists, hand-written for gfx950 (they replace
// rocprim::exclusive_scan and rocprim::segmented_radix_sort_keys, the last rocprim calls of the library):
//   * exclusive scan of per-query counts (u32, optionally clamped to max_nn) into 64-bit offsets;
//   * ascending sort of every query's segment of 64-bit (distance bits, index) keys, in place.
// Segments are short in practice (a radius that holds tens of neighbours), so the sort is a comparator network:
// up to 64 keys in one wavefront's registers (shuffles), up to 4096 in one workgroup's LDS, anything longer in global
// memory by one workgroup.  The network is the NORMALISED bitonic sort -- every comparator orders its pair the same
// way (first step of a merge: i against its mirror in the block, then the usual halvings) -- so a segment is padded
// to a power of two with virtual +inf keys that never move: comparators that would touch one are skipped.
#pragma once

#include "pclhip_internal.hpp"

namespace pclhip {
namespace {

constexpr int SCAN64_BLOCK = 1024;  // elements per workgroup of the two outer passes

// count of query i as the scan sees it
__device__ __forceinline__ unsigned long long scan64_in(const uint32_t* counts, uint32_t n, uint32_t i, uint32_t clamp) {
  if (i >= n) return 0ull;  // the (n + 1)-th element: total
  const uint32_t c = counts[i];
  return (clamp != 0u && c > clamp) ? clamp : c;
}

__global__ __launch_bounds__(256) void scan64_partial_kernel(const uint32_t* __restrict__ counts, uint32_t n, uint32_t clamp,
                                                             unsigned long long* __restrict__ block_sum) {
  __shared__ unsigned long long red[4];
  unsigned long long s = 0;
  const uint32_t base = blockIdx.x * SCAN64_BLOCK;
  for (uint32_t t = threadIdx.x; t < SCAN64_BLOCK; t += 256) s += scan64_in(counts, n, base + t, clamp);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) block_sum[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one workgroup: exclusive scan of the block sums in place
__global__ __launch_bounds__(1024) void scan64_top_kernel(unsigned long long* __restrict__ block_sum, uint32_t nblocks) {
  __shared__ unsigned long long wave_tot[16];
  __shared__ unsigned long long carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < nblocks; b0 += 1024) {
    const uint32_t i = b0 + threadIdx.x;
    const unsigned long long v = i < nblocks ? block_sum[i] : 0ull;
    unsigned long long inc = v;  // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned long long t = __shfl_up(inc, o);
      if (int(threadIdx.x & 63) >= o) inc += t;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = inc;
    __syncthreads();
    unsigned long long before = carry_s;
    for (int w = 0; w < int(threadIdx.x >> 6); ++w) before += wave_tot[w];
    if (i < nblocks) block_sum[i] = before + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = before + inc;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void scan64_apply_kernel(const uint32_t* __restrict__ counts, uint32_t n, uint32_t clamp,
                                                           const unsigned long long* __restrict__ block_sum,
                                                           unsigned long long* __restrict__ out) {
  // 256 threads x 4 consecutive elements: thread-local sums, wave scan, block scan
  __shared__ unsigned long long wave_tot[4];
  const uint32_t base = blockIdx.x * SCAN64_BLOCK + threadIdx.x * 4;
  unsigned long long v[4], s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = scan64_in(counts, n, base + j, clamp);
    s += v[j];
  }
  unsigned long long inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long t = __shfl_up(inc, o);
    if (int(threadIdx.x & 63) >= o) inc += t;
  }
  if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = inc;
  __syncthreads();
  unsigned long long run = block_sum[blockIdx.x] + inc - s;
  for (int w = 0; w < int(threadIdx.x >> 6); ++w) run += wave_tot[w];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (base + j <= n) out[base + j] = run;  // n + 1 outputs: out[n] = total
    run += v[j];
  }
}

// out[i] = sum of min(counts[j], clamp or inf) over j < i, for i = 0 .. n; `block_sum`: ceil((n + 1) / 1024) u64 of scratch
inline void launch_exclusive_scan_u64(hipStream_t s, const uint32_t* counts, uint32_t n, uint32_t clamp,
                                      unsigned long long* block_sum, unsigned long long* out) {
  const uint32_t blocks = (n + 1 + SCAN64_BLOCK - 1) / SCAN64_BLOCK;
  hipLaunchKernelGGL(scan64_partial_kernel, dim3(blocks), dim3(256), 0, s, counts, n, clamp, block_sum);
  hipLaunchKernelGGL(scan64_top_kernel, dim3(1), dim3(1024), 0, s, block_sum, blocks);
  hipLaunchKernelGGL(scan64_apply_kernel, dim3(blocks), dim3(256), 0, s, counts, n, clamp, block_sum, out);
}

// ---- segmented sort -----------------------------------------------------------------------------------
constexpr uint64_t SEG_INF = ~0ull;
constexpr int SEG_LDS_MAX = 4096;

// normalised bitonic network over `len` keys reached through get / set; `tid` of `nthreads` cooperating threads;
// `sync()` orders the steps.  P = len rounded up to a power of two.
template <class Get, class Set, class Sync>
__device__ __forceinline__ void bitonic_network(uint32_t len, uint32_t tid, uint32_t nthreads, Get get, Set set, Sync sync) {
  uint32_t P = 1;
  while (P < len) P <<= 1;
  for (uint32_t k = 2; k <= P; k <<= 1) {
    // first step of the merge of blocks of k: i <-> mirror inside the block
    for (uint32_t t = tid; t < P / 2; t += nthreads) {
      const uint32_t blk = t / (k / 2), off = t % (k / 2);
      const uint32_t a = blk * k + off, b = blk * k + k - 1 - off;
      if (b < len) {
        const uint64_t x = get(a), y = get(b);
        if (y < x) {
          set(a, y);
          set(b, x);
        }
      }
    }
    sync();
    for (uint32_t j = k / 4; j >= 1; j >>= 1) {
      for (uint32_t t = tid; t < P / 2; t += nthreads) {
        const uint32_t a = (t / j) * 2 * j + (t % j), b = a + j;
        if (b < len) {
          const uint64_t x = get(a), y = get(b);
          if (y < x) {
            set(a, y);
            set(b, x);
          }
        }
      }
      sync();
    }
  }
}

// Segments of up to 64 keys: one wavefront each, keys in registers (lane i holds key i, +inf beyond the segment).
// Longer segments are appended to `long_list` for seg_sort_block_kernel.  seg_off: n + 1 offsets (u64), less `base`.
__global__ __launch_bounds__(256) void seg_sort_wave_kernel(uint64_t* __restrict__ keys, const unsigned long long* __restrict__ seg_off,
                                                            unsigned long long base, uint32_t seg_begin, uint32_t seg_end,
                                                            uint32_t* __restrict__ long_list, uint32_t* __restrict__ long_count) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t seg = seg_begin + blockIdx.x * 4u + (threadIdx.x >> 6);
  if (seg >= seg_end) return;
  const unsigned long long b = seg_off[seg] - base;
  const uint32_t len = uint32_t(seg_off[seg + 1] - seg_off[seg]);
  if (len <= 1u) return;
  if (len > 64u) {
    if (lane == 0) long_list[atomicAdd(long_count, 1u)] = seg;
    return;
  }
  uint64_t v = lane < len ? keys[b + lane] : SEG_INF;
  // the same normalised network on lanes: partner by mirror (first step of a merge), then by xor
  uint32_t P = 1;
  while (P < len) P <<= 1;
  for (uint32_t k = 2; k <= P; k <<= 1) {
    {
      const uint32_t partner = (lane & ~(k - 1u)) + (k - 1u - (lane & (k - 1u)));
      const uint64_t o = __shfl(v, int(partner));
      const bool lower = lane < partner;
      v = (lower == (o < v)) ? o : v;  // the lower lane keeps the smaller key
    }
    for (uint32_t j = k / 4; j >= 1; j >>= 1) {
      const uint32_t partner = lane ^ j;
      const uint64_t o = __shfl(v, int(partner));
      const bool lower = lane < partner;
      v = (lower == (o < v)) ? o : v;
    }
  }
  if (lane < len) keys[b + lane] = v;
}

// Long segments: one workgroup each -- in LDS up to SEG_LDS_MAX keys, in global memory beyond.
__global__ __launch_bounds__(1024) void seg_sort_block_kernel(uint64_t* __restrict__ keys, const unsigned long long* __restrict__ seg_off,
                                                              unsigned long long base, const uint32_t* __restrict__ long_list,
                                                              const uint32_t* __restrict__ long_count) {
  __shared__ uint64_t lds[SEG_LDS_MAX];
  const uint32_t count = *long_count;
  for (uint32_t li = blockIdx.x; li < count; li += gridDim.x) {
    const uint32_t seg = long_list[li];
    uint64_t* const k = keys + (seg_off[seg] - base);
    const uint32_t len = uint32_t(seg_off[seg + 1] - seg_off[seg]);
    if (len <= uint32_t(SEG_LDS_MAX)) {
      for (uint32_t t = threadIdx.x; t < len; t += blockDim.x) lds[t] = k[t];
      __syncthreads();
      bitonic_network(len, threadIdx.x, blockDim.x, [&](uint32_t i) { return lds[i]; }, [&](uint32_t i, uint64_t x) { lds[i] = x; },
                      [] { __syncthreads(); });
      for (uint32_t t = threadIdx.x; t < len; t += blockDim.x) k[t] = lds[t];
      __syncthreads();
    } else {
      bitonic_network(len, threadIdx.x, blockDim.x, [&](uint32_t i) { return k[i]; }, [&](uint32_t i, uint64_t x) { k[i] = x; },
                      [] {
                        __threadfence_block();
                        __syncthreads();
                      });
    }
  }
}

// sort every segment [seg_begin, seg_end) of `keys` (keys of segment i at seg_off[i] - base) ascending, in place;
// long_list: (seg_end - seg_begin) u32 of scratch, long_count: one u32 of scratch
inline void launch_segmented_sort_u64(hipStream_t s, int num_cus, uint64_t* keys, const unsigned long long* seg_off,
                                      unsigned long long base, uint32_t seg_begin, uint32_t seg_end, uint32_t* long_list,
                                      uint32_t* long_count) {
  if (seg_end <= seg_begin) return;
  (void)hipMemsetAsync(long_count, 0, sizeof(uint32_t), s);
  const uint32_t nseg = seg_end - seg_begin;
  hipLaunchKernelGGL(seg_sort_wave_kernel, dim3((nseg + 3) / 4), dim3(256), 0, s, keys, seg_off, base, seg_begin, seg_end,
                     long_list, long_count);
  hipLaunchKernelGGL(seg_sort_block_kernel, dim3(unsigned(num_cus) * 2u), dim3(1024), 0, s, keys, seg_off, base, long_list,
                     long_count);
}

}  // namespace
}  // namespace pclhip
