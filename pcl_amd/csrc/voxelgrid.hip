// voxelgrid.hip -- pcl::VoxelGrid<pcl::PointXYZ>::applyFilter on the GPU
// (filters/include/pcl/filters/impl/voxel_grid.hpp:597-814): bounding box -> int32 voxel id per
// point (same float expressions, :713-718) -> stable radix sort of (voxel id, point index) (hand-written, below) -> run
// boundaries -> per-voxel centroid (float sum / count, common/include/pcl/common/impl/
// accumulators.hpp:68-85) in ascending voxel id order.  Within a voxel the points are summed in
// ascending input index order (the stable sort fixes what the reference's spreadsort leaves
// unspecified), so centroids are bit-identical to the CPU oracle.
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>

#include <algorithm>
#include <cfloat>
#include <cmath>

#include "pclhip_internal.hpp"
#include "device_scan.hpp"

namespace pclhip {
namespace {

__device__ __forceinline__ const float* rec(const void* base, size_t stride, uint64_t i) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + i * stride);
}

// The pass-through filter in front of the grid (setFilterFieldName / setFilterLimits / setFilterLimitsNegative,
// voxel_grid.h:440-476): `limits` = 0 off; bit 0 on, bit 1 negative, bits 8..15 = 1 + position of the field in the
// record counted in floats (0: the z coordinate, as the flag meant before it carried a field).
__device__ __forceinline__ int limit_field(int limits) {
  const int f = (limits >> 8) & 0xFF;
  return f ? f - 1 : 2;
}

constexpr int VG_BATCH = 4;   // records a thread asks for at a time in the grid-stride passes over the cloud

// getMinMax3D with the optional field filter (voxel_grid.hpp:513-590; limits cast to float, :615)
__global__ __launch_bounds__(256) void vg_minmax_kernel(const void* pts, size_t stride, uint64_t n, int has_limits,
                                                        float fmin_, float fmax_, float* partial) {
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  // VG_BATCH records per trip of the grid-stride loop, their loads in flight together (one record per trip is one memory
  // round trip per trip: twenty in a row at 10M points); slots past the end re-read the trip's first record and are skipped
  const uint64_t G = uint64_t(gridDim.x) * blockDim.x;
  for (uint64_t i0 = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x; i0 < n; i0 += VG_BATCH * G) {
    float xs[VG_BATCH], ys[VG_BATCH], zs[VG_BATCH], vs[VG_BATCH];
#pragma unroll
    for (int u = 0; u < VG_BATCH; ++u) {
      const uint64_t i = i0 + uint64_t(u) * G;
      const float* p = rec(pts, stride, i < n ? i : i0);
      xs[u] = p[0]; ys[u] = p[1]; zs[u] = p[2];
    }
    if (has_limits) {
#pragma unroll
      for (int u = 0; u < VG_BATCH; ++u) {
        const uint64_t i = i0 + uint64_t(u) * G;
        vs[u] = rec(pts, stride, i < n ? i : i0)[limit_field(has_limits)];
      }
    }
#pragma unroll
    for (int u = 0; u < VG_BATCH; ++u) {
      if (i0 + uint64_t(u) * G >= n) continue;
      const float x = xs[u], y = ys[u], z = zs[u];
      if (has_limits) {  // voxel_grid.hpp:513-590: the field's value against the limits cast to float
        const float v = vs[u];
        if (has_limits & 2) {  // filter_limit_negative_: points INSIDE the interval are cut
          if ((v < fmax_) && (v > fmin_)) continue;
        } else if ((v > fmax_) || (v < fmin_)) {
          continue;
        }
      }
      if (!(isfinite(x) && isfinite(y) && isfinite(z))) continue;
      lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
      hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
    }
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
                                                     double lim_min, double lim_max, uint32_t* keys,
                                                     unsigned int* n_valid, int sort_bits) {
  unsigned int mine = 0;
  const uint64_t G = uint64_t(gridDim.x) * blockDim.x;
  for (uint64_t i0 = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x; i0 < n; i0 += VG_BATCH * G) {   // as vg_minmax_kernel
    float xs[VG_BATCH], ys[VG_BATCH], zs[VG_BATCH], vs[VG_BATCH];
#pragma unroll
    for (int u = 0; u < VG_BATCH; ++u) {
      const uint64_t i = i0 + uint64_t(u) * G;
      const float* p = rec(pts, stride, i < n ? i : i0);
      xs[u] = p[0]; ys[u] = p[1]; zs[u] = p[2];
    }
    if (has_limits) {
#pragma unroll
      for (int u = 0; u < VG_BATCH; ++u) {
        const uint64_t i = i0 + uint64_t(u) * G;
        vs[u] = rec(pts, stride, i < n ? i : i0)[limit_field(has_limits)];
      }
    }
#pragma unroll
    for (int u = 0; u < VG_BATCH; ++u) {
      const uint64_t i = i0 + uint64_t(u) * G;
      if (i >= n) continue;
      const float x = xs[u], y = ys[u], z = zs[u];
      bool ok = isfinite(x) && isfinite(y) && isfinite(z);
      if (ok && has_limits) {  // :684-695: double limits against the float value
        const double v = double(vs[u]);
        ok = (has_limits & 2) ? !((v < lim_max) && (v > lim_min)) : !((v > lim_max) || (v < lim_min));
      }
      uint32_t key = sort_bits >= 32 ? 0xFFFFFFFFu : ((1u << sort_bits) - 1u);  // rejected: after every voxel id
      if (ok) {  // :713-718
        const int i0v = int(floorf(__fmul_rn(x, g.inv[0])) - float(g.min_b[0]));
        const int i1v = int(floorf(__fmul_rn(y, g.inv[1])) - float(g.min_b[1]));
        const int i2v = int(floorf(__fmul_rn(z, g.inv[2])) - float(g.min_b[2]));
        key = uint32_t(i0v * g.mul[0] + i1v * g.mul[1] + i2v * g.mul[2]);
        ++mine;
      }
      keys[i] = key;
    }
  }
  // one atomic per block of a grid that is sized to the machine, not to the cloud (39k single-address atomics -- one
  // per 256 points -- took 0.4 ms of this kernel's 0.45 at 10M points; 4096 blocks still 28 us of 64: two blocks per CU now)
  __shared__ unsigned int blk;
  if (threadIdx.x == 0) blk = 0;
  __syncthreads();
  if (mine) atomicAdd(&blk, mine);
  __syncthreads();
  if (threadIdx.x == 0 && blk) atomicAdd(n_valid, blk);
}

// Runs of equal keys in the sorted order (the number of valid points nv <= n is read from device memory: the host does
// not wait for it before these launches).  Rounds 1-3 wrote a head flag per point, scanned all n flags and scattered
// the run starts from both arrays (140 us of a 0.92 ms filter at 10M points); only the run STARTS are wanted, so:
//   vg_runcount_kernel   heads per block of 4096 sorted keys                      (reads the keys)
//   launch_scan_u32      over the block counts (a few thousand numbers; its total = number of runs)
//   vg_runstart_kernel   every block finds its heads again, ranks them in index order (ballots inside a wavefront,
//                        a prefix over the block's 64 (row, wave) cells) and writes run_start[block prefix + rank]
constexpr int VG_RUN_BLOCK = 4096, VG_RUN_THREADS = 256, VG_RUN_ROWS = VG_RUN_BLOCK / VG_RUN_THREADS;

// A thread's sixteen keys and their predecessors, every load in flight before the first comparison (a load inside
// `j < nv && ...` is followed by its own wait: sixteen memory round trips one after the other).  Rows past the end read key 0.
__device__ __forceinline__ void vg_load_pairs(const uint32_t* __restrict__ keys, uint32_t base, uint32_t nv,
                                              uint32_t (&kc)[VG_RUN_ROWS], uint32_t (&kp)[VG_RUN_ROWS]) {
#pragma unroll
  for (int e = 0; e < VG_RUN_ROWS; ++e) {
    const uint32_t j = base + uint32_t(e) * VG_RUN_THREADS + threadIdx.x;
    kc[e] = keys[j < nv ? j : 0u];
    kp[e] = keys[(j < nv && j > 0u) ? j - 1u : 0u];
  }
}
__device__ __forceinline__ bool vg_is_head(uint32_t key, uint32_t prev, uint32_t j, uint32_t nv) {
  return j < nv && (j == 0u || key != prev);
}

__global__ __launch_bounds__(VG_RUN_THREADS) void vg_runcount_kernel(const uint32_t* __restrict__ keys,
                                                                     const unsigned int* __restrict__ nv_dev,
                                                                     uint32_t* __restrict__ block_heads) {
  __shared__ uint32_t total;
  if (threadIdx.x == 0) total = 0u;
  __syncthreads();
  const uint32_t nv = *nv_dev;
  const uint32_t base = blockIdx.x * uint32_t(VG_RUN_BLOCK);
  uint32_t mine = 0;
  uint32_t kc[VG_RUN_ROWS], kp[VG_RUN_ROWS];
  vg_load_pairs(keys, base, nv, kc, kp);
#pragma unroll
  for (int e = 0; e < VG_RUN_ROWS; ++e) mine += vg_is_head(kc[e], kp[e], base + uint32_t(e) * VG_RUN_THREADS + threadIdx.x, nv) ? 1u : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
  if ((threadIdx.x & 63u) == 0u && mine) atomicAdd(&total, mine);
  __syncthreads();
  if (threadIdx.x == 0) block_heads[blockIdx.x] = total;
}

// run_start[r] = first sorted position of run r; run_start[number of runs] = nv (end sentinel)
__global__ __launch_bounds__(VG_RUN_THREADS) void vg_runstart_kernel(const uint32_t* __restrict__ keys,
                                                                     const unsigned int* __restrict__ nv_dev,
                                                                     const uint32_t* __restrict__ block_first,
                                                                     uint32_t* __restrict__ run_start) {
  constexpr int WAVES = VG_RUN_THREADS / 64;
  __shared__ uint32_t cell[VG_RUN_ROWS * WAVES];   // heads per (row, wave), then their exclusive prefix
  const uint32_t nv = *nv_dev;
  const uint32_t base = blockIdx.x * uint32_t(VG_RUN_BLOCK);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  uint32_t rank[VG_RUN_ROWS];
  uint32_t heads = 0;   // bit e: this thread's element of row e starts a run
  uint32_t kc[VG_RUN_ROWS], kp[VG_RUN_ROWS];
  vg_load_pairs(keys, base, nv, kc, kp);
#pragma unroll
  for (int e = 0; e < VG_RUN_ROWS; ++e) {
    const bool h = vg_is_head(kc[e], kp[e], base + uint32_t(e) * VG_RUN_THREADS + threadIdx.x, nv);
    const unsigned long long b = __builtin_amdgcn_ballot_w64(h);
    rank[e] = uint32_t(__builtin_popcountll(b & below));
    heads |= h ? (1u << e) : 0u;
    if (lane == 0) cell[e * WAVES + int(wave)] = uint32_t(__builtin_popcountll(b));
  }
  __syncthreads();
  if (threadIdx.x < 64u) {   // exclusive prefix over the 64 cells (position order), one wavefront
    const uint32_t own = cell[lane];
    uint32_t v = own;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(v, o);
      if (lane >= uint32_t(o)) v += t;
    }
    cell[lane] = v - own;
  }
  __syncthreads();
  const uint32_t first = block_first[blockIdx.x];
#pragma unroll
  for (int e = 0; e < VG_RUN_ROWS; ++e) {
    const uint32_t j = base + uint32_t(e) * VG_RUN_THREADS + threadIdx.x;
    const uint32_t r = first + cell[e * WAVES + int(wave)] + rank[e];
    if (heads & (1u << e)) run_start[r] = j;
    if (nv != 0u && j == nv - 1u) run_start[r + ((heads >> e) & 1u)] = nv;  // end sentinel: behind the last run
  }
}

__global__ void vg_counts_kernel(const unsigned int* __restrict__ nv_dev, const uint32_t* __restrict__ tot, uint32_t* __restrict__ host) {
  host[0] = *nv_dev;
  host[1] = tot[0];
}

__global__ void vg_keep_kernel(const uint32_t* run_start, uint32_t nruns, uint32_t min_pts, uint32_t* keep) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < nruns) keep[r] = ((run_start[r + 1] - run_start[r]) >= min_pts) ? 1u : 0u;
}

// ---- stable LSD radix sort of (u32 key, u32 value) pairs, 8 bits per pass --------------------------------------
// Per pass: digit histogram of every 4096-key block (rs_hist_kernel, digit-major), the scan of every digit's row over
// the blocks (rs_rowscan_kernel: first output slot of (digit, block) relative to the digit's first), and the scatter:
// a wavefront owns 1024 consecutive keys as 16 rows of 64; lanes with equal digits find each other with eight
// ballots, the lowest of them bumps the wave's running count of that digit (LDS, no atomics: one leader per digit
// and row), so every key gets its rank among the equal digits before it -- index order, hence stable.
// (Digits of 10 bits -- two passes instead of three for the 20 key bits of a 0.01 grid over the bench's cloud -- were
// measured: 1024 bins scatter into shorter segments and need 20 KB of LDS per block, 1.10 against 0.93 ms per cloud.)
constexpr int RS_KPB = 4096, RS_THREADS = 256, RS_ROWS = RS_KPB / RS_THREADS;  // 16 rows per thread

__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n, int shift,
                                                             uint32_t nblocks, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t base = blockIdx.x * uint32_t(RS_KPB);
  uint32_t kv[RS_ROWS];   // all sixteen loads in flight before the first use (rows past the end read key 0)
#pragma unroll
  for (int r = 0; r < RS_ROWS; ++r) {
    const uint32_t i = base + uint32_t(r) * RS_THREADS + threadIdx.x;
    kv[r] = keys[i < n ? i : 0u];
  }
#pragma unroll
  for (int r = 0; r < RS_ROWS; ++r) {
    const uint32_t i = base + uint32_t(r) * RS_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&h[(kv[r] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[size_t(threadIdx.x) * nblocks + blockIdx.x] = h[threadIdx.x];
}

// Offsets of one sorting pass in ONE launch: workgroup d scans row d of the digit-major table in place of the generic
// three-launch scan of all 256 x nblocks counters (first[d][b] = keys of digit d in the blocks before b) and leaves the
// row total; the scatter kernel adds the totals of the smaller digits itself (256 values, scanned in LDS).
__global__ __launch_bounds__(256) void rs_rowscan_kernel(const uint32_t* __restrict__ hist, uint32_t nblocks,
                                                         uint32_t* __restrict__ first, uint32_t* __restrict__ rowtot) {
  const uint32_t* row = hist + size_t(blockIdx.x) * nblocks;
  uint32_t* out = first + size_t(blockIdx.x) * nblocks;
  const uint32_t per = (nblocks + 255u) / 256u;
  const uint32_t b0 = threadIdx.x * per, b1 = (b0 + per < nblocks) ? b0 + per : nblocks;
  uint32_t sum = 0;
  for (uint32_t b = b0; b < b1; ++b) sum += row[b];
  // exclusive scan of the 256 chunk sums: wave scans, then the four wave totals
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o);
    if (lane >= uint32_t(o)) inc += t;
  }
  __shared__ uint32_t wtot[4];
  if (lane == 63u) wtot[wave] = inc;
  __syncthreads();
  uint32_t before = inc - sum;
  for (uint32_t w = 0; w < wave; ++w) before += wtot[w];
  for (uint32_t b = b0; b < b1; ++b) {
    const uint32_t v = row[b];
    out[b] = before;
    before += v;
  }
  if (threadIdx.x == 255) rowtot[blockIdx.x] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
}

__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                const uint32_t* __restrict__ vals_in, uint32_t n, int shift,
                                                                uint32_t nblocks, const uint32_t* __restrict__ first,
                                                                const uint32_t* __restrict__ rowtot,
                                                                uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
  constexpr int WAVES = RS_THREADS / 64;
  static_assert(RS_THREADS == 256, "one thread per digit");
  __shared__ uint32_t wcnt[WAVES][256];
  __shared__ uint32_t goff[256];
  __shared__ uint32_t lstart[256];
  __shared__ uint32_t dtot[WAVES];
  __shared__ uint32_t skey[RS_KPB], sval[RS_KPB];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < WAVES * 256; i += RS_THREADS) (&wcnt[0][0])[i] = 0u;
  {  // keys of smaller digits (exclusive scan of the 256 row totals) + keys of this digit in earlier blocks
    const uint32_t tot = rowtot[threadIdx.x];
    uint32_t inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o);
      if (lane >= uint32_t(o)) inc += t;
    }
    if (lane == 63u) dtot[wave] = inc;
    __syncthreads();
    uint32_t base = inc - tot;
    for (uint32_t w = 0; w < wave; ++w) base += dtot[w];
    goff[threadIdx.x] = base + first[size_t(threadIdx.x) * nblocks + blockIdx.x];
  }
  __syncthreads();
  const unsigned long long below = (1ull << lane) - 1ull;
  const uint32_t base = blockIdx.x * uint32_t(RS_KPB) + wave * uint32_t(RS_ROWS * 64);
  uint32_t key[RS_ROWS], val[RS_ROWS], rk[RS_ROWS];
  // keys and values of all sixteen rows first, every load in flight together (the ranking below has a wave barrier per row,
  // and a load asked for inside it was waited for inside it); rows past the end read element 0 and are masked below
#pragma unroll
  for (int r = 0; r < RS_ROWS; ++r) {
    const uint32_t i = base + uint32_t(r) * 64u + lane;
    key[r] = keys_in[i < n ? i : 0u];
  }
  if (vals_in) {
#pragma unroll
    for (int r = 0; r < RS_ROWS; ++r) {
      const uint32_t i = base + uint32_t(r) * 64u + lane;
      val[r] = vals_in[i < n ? i : 0u];
    }
  } else {   // first pass: the value is the point index
#pragma unroll
    for (int r = 0; r < RS_ROWS; ++r) val[r] = base + uint32_t(r) * 64u + lane;
  }
#pragma unroll
  for (int r = 0; r < RS_ROWS; ++r) {
    const uint32_t i = base + uint32_t(r) * 64u + lane;
    const bool ok = i < n;
    if (!ok) key[r] = val[r] = 0u;
    const uint32_t d = (key[r] >> shift) & 255u;
    unsigned long long peers = __builtin_amdgcn_ballot_w64(ok);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long m = __builtin_amdgcn_ballot_w64((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    uint32_t before = 0;
    if (ok) {
      const int leader = __builtin_ctzll(peers);
      if (int(lane) == leader) {
        before = wcnt[wave][d];
        wcnt[wave][d] = before + uint32_t(__builtin_popcountll(peers));
      }
    }
    // every lane takes part in the shuffle; lanes without a key read their own (unused) value
    const int src = ok ? __builtin_ctzll(peers) : int(lane);
    before = __shfl(before, src);
    rk[r] = ok ? ((d << 16) | (before + uint32_t(__builtin_popcountll(peers & below)))) : 0xFFFFFFFFu;
    __builtin_amdgcn_wave_barrier();  // the next row's leader reads what this row's leader wrote
  }
  __syncthreads();
  // The block's pairs go out through LDS in their sorted order: a lane that writes its own pair to its final slot sends 64
  // four-byte stores to 64 places per instruction (a digit holds 16 of the block's keys on average: 1.9 TB/s of pairs at
  // 10M points); staged, consecutive lanes write consecutive slots of a digit's segment.
  // digit d of this block: first slot in the staged order (digits ascending, waves in order inside a digit)
  {
    const uint32_t d = threadIdx.x;
    uint32_t c[WAVES], tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      c[w] = wcnt[w][d];
      tot += c[w];
    }
    uint32_t inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o);
      if (lane >= uint32_t(o)) inc += t;
    }
    if (lane == 63u) dtot[wave] = inc;   // (its first use ended at the barrier behind goff)
    __syncthreads();
    uint32_t run = inc - tot;
    for (uint32_t w = 0; w < wave; ++w) run += dtot[w];
    lstart[d] = run;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      wcnt[w][d] = run;
      run += c[w];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_ROWS; ++r) {
    if (rk[r] != 0xFFFFFFFFu) {
      const uint32_t d = rk[r] >> 16, lp = wcnt[wave][d] + (rk[r] & 0xFFFFu);
      skey[lp] = key[r];
      sval[lp] = val[r];
    }
  }
  __syncthreads();
  const uint32_t block_first = blockIdx.x * uint32_t(RS_KPB);
  const uint32_t live = n - block_first < uint32_t(RS_KPB) ? n - block_first : uint32_t(RS_KPB);   // pairs of this block
#pragma unroll
  for (int r = 0; r < RS_ROWS; ++r) {
    const uint32_t j = uint32_t(r) * RS_THREADS + threadIdx.x;
    if (j < live) {
      const uint32_t k = skey[j], d = (k >> shift) & 255u, pos = goff[d] + (j - lstart[d]);
      keys_out[pos] = k;
      vals_out[pos] = sval[j];
    }
  }
}

// setSaveLeafLayout (impl/voxel_grid.hpp:752-787): layout[voxel id] = position of its centroid in the output
__global__ void vg_layout_kernel(const uint32_t* __restrict__ keys_sorted, const uint32_t* __restrict__ run_start,
                                 const uint32_t* __restrict__ keep, const uint32_t* __restrict__ keep_scan, uint32_t nruns,
                                 int32_t* __restrict__ layout) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < nruns && keep[r]) layout[keys_sorted[run_start[r]]] = int32_t(keep_scan[r]);
}

// one thread per voxel: sequential float accumulation in sorted (= ascending input index) order.
// Output records of `ostride` bytes: x y z 1 at +0; with normals (noff != 0, pcl::PointNormal layout) and
// downsample_all_data the CentroidPoint accumulators of common/include/pcl/common/impl/accumulators.hpp:68-127:
// normal = normalized sum (Eigen's Vector4f::normalized: v / sqrt(sum of squares); 0/0 -> NaN like Eigen >= 3.3),
// curvature = sum / n.  Without downsample_all_data only the coordinates are averaged (voxel_grid.hpp:790-799),
// every other field of the output record keeps its default (0).
__global__ __launch_bounds__(256) void vg_centroid_kernel(const void* pts, size_t stride, const uint32_t* vals,
                                                          const uint32_t* run_start, const uint32_t* keep,
                                                          const uint32_t* keep_scan, uint32_t nruns, void* out,
                                                          size_t ostride, size_t noff, int all_data) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nruns || !keep[r]) return;
  const uint32_t b = run_start[r], e = run_start[r + 1];
  float sx = 0.0f, sy = 0.0f, sz = 0.0f, nx = 0.0f, ny = 0.0f, nz = 0.0f, cv = 0.0f;
  const bool with_n = noff != 0 && all_data;
  for (uint32_t j = b; j < e; ++j) {
    const float* p = rec(pts, stride, vals[j]);
    sx = __fadd_rn(sx, p[0]);
    sy = __fadd_rn(sy, p[1]);
    sz = __fadd_rn(sz, p[2]);
    if (with_n) {
      const float* q = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + noff);
      nx = __fadd_rn(nx, q[0]);
      ny = __fadd_rn(ny, q[1]);
      nz = __fadd_rn(nz, q[2]);
      cv = __fadd_rn(cv, q[4]);
    }
  }
  const float cnt = float(e - b);
  float* o = reinterpret_cast<float*>(reinterpret_cast<char*>(out) + size_t(keep_scan[r]) * ostride);  // keep_scan: exclusive scan of keep
  o[0] = __fdiv_rn(sx, cnt);
  o[1] = __fdiv_rn(sy, cnt);
  o[2] = __fdiv_rn(sz, cnt);
  if (ostride >= 16) o[3] = 1.0f;
  if (noff != 0) {
    float* q = reinterpret_cast<float*>(reinterpret_cast<char*>(o) + noff);
    float a = 0.0f, bb = 0.0f, c = 0.0f, k = 0.0f;
    if (with_n) {
      const float len = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(nx, nx), __fmul_rn(ny, ny)), __fmul_rn(nz, nz)));
      a = __fdiv_rn(nx, len);
      bb = __fdiv_rn(ny, len);
      c = __fdiv_rn(nz, len);
      k = __fdiv_rn(cv, cnt);
    }
    q[0] = a; q[1] = bb; q[2] = c; q[3] = 0.0f;
    q[4] = k; q[5] = 0.0f; q[6] = 0.0f; q[7] = 0.0f;
  }
}

// Four runs per wavefront, one per row of 16 lanes: a row gathers 16 points of its run at a time and every lane of the
// row adds them up in index order from row broadcasts (ds_bpermute) -- the same sequential float sums, a quarter of the
// broadcast + add instructions per point of the wave-per-run form.
__global__ __launch_bounds__(256) void vg_centroid_row_kernel(const void* pts, size_t stride, const uint32_t* vals,
                                                              const uint32_t* run_start, const uint32_t* keep,
                                                              const uint32_t* keep_scan, uint32_t nruns, void* out,
                                                              size_t ostride, size_t noff, int all_data) {
  const uint32_t lane = threadIdx.x & 63u, sub = lane & 15u, rowbase = lane & ~15u;
  const uint32_t row = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, nrows = (gridDim.x * blockDim.x) >> 4;
  const bool with_n = noff != 0 && all_data;
  const uint32_t rounds = (nruns + nrows - 1) / nrows;  // every row of a wave takes part in every shuffle
  for (uint32_t it = 0; it < rounds; ++it) {
    const uint32_t r = it * nrows + row;
    const bool live = r < nruns && keep[r] != 0;
    const uint32_t b = live ? run_start[r] : 0u, e = live ? run_start[r + 1] : 0u;
    const uint32_t len = e - b;
    uint32_t longest = len;  // over the four rows of the wave
    longest = max(longest, uint32_t(__shfl_xor(int(longest), 16)));
    longest = max(longest, uint32_t(__shfl_xor(int(longest), 32)));
    float sx = 0.0f, sy = 0.0f, sz = 0.0f, nx = 0.0f, ny = 0.0f, nz = 0.0f, cv = 0.0f;
    // The gather is two dependent loads per chunk of 16 points (the sorted index, then the record it names).  They run two
    // chunks deep: while chunk c is added up, the records of chunk c + 16 and the indices of chunk c + 32 are on their way
    // (one after the other they were two memory round trips per chunk and row: 199 us at 10M points).  Slots past the
    // run's end re-read its first element (always there: position 0 when the row has no run) and never enter a sum.
    const auto slot = [&](uint32_t c) { return (c + sub < len) ? b + c + sub : b; };
    float px, py, pz, qx = 0.0f, qy = 0.0f, qz = 0.0f, qc = 0.0f;
    uint32_t idx_next;
    {
      const float* p = rec(pts, stride, vals[slot(0u)]);
      idx_next = vals[slot(16u)];
      px = p[0]; py = p[1]; pz = p[2];
      if (with_n) {
        const float* q = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + noff);
        qx = q[0]; qy = q[1]; qz = q[2]; qc = q[4];
      }
    }
    for (uint32_t c = 0; c < longest; c += 16u) {
      // chunk c + 16's records, chunk c + 32's indices
      const float* pn = rec(pts, stride, idx_next);
      const float npx = pn[0], npy = pn[1], npz = pn[2];
      float nqx = 0.0f, nqy = 0.0f, nqz = 0.0f, nqc = 0.0f;
      if (with_n) {
        const float* q = reinterpret_cast<const float*>(reinterpret_cast<const char*>(pn) + noff);
        nqx = q[0]; nqy = q[1]; nqz = q[2]; nqc = q[4];
      }
      idx_next = vals[slot(c + 32u)];
#pragma unroll
      for (uint32_t j = 0; j < 16u; ++j) {
        const int src = int(rowbase + j);
        const float vx = __shfl(px, src), vy = __shfl(py, src), vz = __shfl(pz, src);
        const bool on = c + j < len;  // the same for the 16 lanes of a row
        sx = on ? __fadd_rn(sx, vx) : sx;
        sy = on ? __fadd_rn(sy, vy) : sy;
        sz = on ? __fadd_rn(sz, vz) : sz;
        if (with_n) {  // wave-uniform
          const float wx = __shfl(qx, src), wy = __shfl(qy, src), wz = __shfl(qz, src), wc = __shfl(qc, src);
          nx = on ? __fadd_rn(nx, wx) : nx;
          ny = on ? __fadd_rn(ny, wy) : ny;
          nz = on ? __fadd_rn(nz, wz) : nz;
          cv = on ? __fadd_rn(cv, wc) : cv;
        }
      }
      px = npx; py = npy; pz = npz;
      qx = nqx; qy = nqy; qz = nqz; qc = nqc;
    }
    if (live && sub == 0u) {
      const float cnt = float(len);
      float* o = reinterpret_cast<float*>(reinterpret_cast<char*>(out) + size_t(keep_scan[r]) * ostride);
      o[0] = __fdiv_rn(sx, cnt);
      o[1] = __fdiv_rn(sy, cnt);
      o[2] = __fdiv_rn(sz, cnt);
      if (ostride >= 16) o[3] = 1.0f;
      if (noff != 0) {
        float* q = reinterpret_cast<float*>(reinterpret_cast<char*>(o) + noff);
        float a = 0.0f, bb = 0.0f, cc = 0.0f, k = 0.0f;
        if (with_n) {
          const float l = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(nx, nx), __fmul_rn(ny, ny)), __fmul_rn(nz, nz)));
          a = __fdiv_rn(nx, l);
          bb = __fdiv_rn(ny, l);
          cc = __fdiv_rn(nz, l);
          k = __fdiv_rn(cv, cnt);
        }
        q[0] = a; q[1] = bb; q[2] = cc; q[3] = 0.0f;
        q[4] = k; q[5] = 0.0f; q[6] = 0.0f; q[7] = 0.0f;
      }
    }
  }
}

}  // namespace
void preload_voxelgrid_kernels() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(vg_key_kernel));
}
}  // namespace pclhip

using namespace pclhip;

extern "C" pclhip_status pclhip_voxelgrid(pclhip_ctx* ctx, const void* points, size_t stride, uint64_t n,
                                          const float leaf[3], uint32_t min_points_per_voxel, int has_z_limits,
                                          double z_min, double z_max, void* out_xyzw, uint64_t* out_n) {
  return pclhip_voxelgrid_ex(ctx, points, stride, n, leaf, min_points_per_voxel, has_z_limits, z_min, z_max, 1, 0, out_xyzw,
                             16, out_n);
}

extern "C" pclhip_status pclhip_voxelgrid_ex(pclhip_ctx* ctx, const void* points, size_t stride, uint64_t n,
                                             const float leaf[3], uint32_t min_points_per_voxel, int has_z_limits,
                                             double z_min, double z_max, int downsample_all_data, size_t normals_offset,
                                             void* out, size_t out_stride, uint64_t* out_n) {
  return pclhip_voxelgrid_ex2(ctx, points, stride, n, leaf, min_points_per_voxel, has_z_limits, z_min, z_max, downsample_all_data,
                              normals_offset, out, out_stride, out_n, nullptr, 0, nullptr);
}

extern "C" pclhip_status pclhip_voxelgrid_ex2(pclhip_ctx* ctx, const void* points, size_t stride, uint64_t n,
                                              const float leaf[3], uint32_t min_points_per_voxel, int has_z_limits,
                                              double z_min, double z_max, int downsample_all_data, size_t normals_offset,
                                              void* out, size_t out_stride, uint64_t* out_n, int32_t* leaf_layout,
                                              uint64_t leaf_layout_capacity, pclhip_voxelgrid_dims* dims) {
  if (!ctx || !leaf || !out_n) return PCLHIP_ERR_INVALID;
  *out_n = 0;
  if (dims) std::memset(dims, 0, sizeof *dims);
  const bool dims_only = out == nullptr && dims != nullptr;   // pclhip_voxelgrid_grid
  PCLHIP_REQUIRE(ctx, stride >= 12 && stride % 4 == 0, "stride must be a multiple of 4 and >= 12 bytes");
  PCLHIP_REQUIRE(ctx, out_stride >= 12 && out_stride % 4 == 0, "output stride must be a multiple of 4 and >= 12 bytes");
  PCLHIP_REQUIRE(ctx, normals_offset == 0 || (normals_offset % 4 == 0 && normals_offset + 32 <= stride &&
                                              normals_offset + 32 <= out_stride && normals_offset >= 12),
                 "normals_offset: the 8 floats (normal[4], curvature, pad[3]) must fit both record layouts");
  PCLHIP_REQUIRE(ctx, n < 0x7FFFFFFFull, "cloud too large for int32 indices");
  PCLHIP_REQUIRE(ctx, leaf[0] > 0 && leaf[1] > 0 && leaf[2] > 0, "leaf size must be positive");
  PCLHIP_REQUIRE(ctx, has_z_limits == 0 || ((has_z_limits & 1) && (has_z_limits & ~0xFF03) == 0 &&
                                           size_t(((has_z_limits >> 8) & 0xFF) ? ((has_z_limits >> 8) & 0xFF) : 3) * 4 <= stride),
                 "filter limits: bit 0 on, bit 1 negative, bits 8..15 = 1 + float position of the field inside the record");
  if (n == 0) return PCLHIP_OK;
  PCLHIP_REQUIRE(ctx, points && (out || dims_only), "null buffer");
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  dev_reserve_for_points(ctx, n);
  hipStream_t s = ctx->stream;
  struct Guard {
    pclhip_ctx* ctx = nullptr;
    std::vector<void*> p;
    ~Guard() {
      for (void* q : p)
        if (q) (void)dev_free(ctx, q);
    }
  } guard;
  guard.ctx = ctx;
  const void* dp = nullptr;
  void* owned = nullptr;
  pclhip_status st = to_device(ctx, points, size_t(n) * stride, &dp, &owned);
  if (st != PCLHIP_OK) return st;
  guard.p.push_back(owned);

  // One scratch block for everything (the context keeps it between calls: no allocation on a warm context).
  auto align = [](size_t x) { return (x + 255) & ~size_t(255); };
  const int nb = int(std::min<uint64_t>((n + 2047) / 2048, 1024));
  const uint32_t sort_blocks = uint32_t((n + RS_KPB - 1) / RS_KPB);
  const size_t table = size_t(256) * sort_blocks;                       // digit histogram of one sorting pass
  const size_t scan_blocks = (std::max<size_t>(table, n) + SC_BLOCK - 1) / SC_BLOCK;
  const size_t o_k0 = 0;
  const size_t o_k1 = o_k0 + align(n * sizeof(uint32_t));
  const size_t o_v0 = o_k1 + align(n * sizeof(uint32_t));
  const size_t o_v1 = o_v0 + align(n * sizeof(uint32_t));
  const size_t o_head = o_v1 + align(n * sizeof(uint32_t));     // head, later keep
  const size_t o_scan = o_head + align(n * sizeof(uint32_t));   // scan, later keep_scan
  const size_t o_runs = o_scan + align(n * sizeof(uint32_t));   // run_start [nruns + 1]
  const size_t o_cnt = o_runs + align((n + 1) * sizeof(uint32_t));
  const size_t o_hist = o_cnt + align(sizeof(unsigned int));
  const size_t o_first = o_hist + align(table * sizeof(uint32_t));
  const size_t o_sp = o_first + align(table * sizeof(uint32_t));
  const size_t o_tot = o_sp + align(scan_blocks * sizeof(uint2));
  const size_t o_rowtot = o_tot + align(4 * sizeof(uint32_t));
  const size_t total_bytes = o_rowtot + align(256 * sizeof(uint32_t));
  st = ensure_scratch(ctx, total_bytes);
  if (st != PCLHIP_OK) return st;
  char* base = static_cast<char*>(ctx->scratch);
  uint32_t* k0 = reinterpret_cast<uint32_t*>(base + o_k0);
  uint32_t* k1 = reinterpret_cast<uint32_t*>(base + o_k1);
  uint32_t* v0 = reinterpret_cast<uint32_t*>(base + o_v0);
  uint32_t* v1 = reinterpret_cast<uint32_t*>(base + o_v1);
  uint32_t* head = reinterpret_cast<uint32_t*>(base + o_head);
  uint32_t* scan = reinterpret_cast<uint32_t*>(base + o_scan);
  uint32_t* run_start = reinterpret_cast<uint32_t*>(base + o_runs);
  unsigned int* d_cnt = reinterpret_cast<unsigned int*>(base + o_cnt);
  uint32_t* hist = reinterpret_cast<uint32_t*>(base + o_hist);
  uint32_t* first = reinterpret_cast<uint32_t*>(base + o_first);
  uint2* sc_partial = reinterpret_cast<uint2*>(base + o_sp);
  uint32_t* tot = reinterpret_cast<uint32_t*>(base + o_tot);
  uint32_t* rowtot = reinterpret_cast<uint32_t*>(base + o_rowtot);
  // exclusive scan of a[0..m) into out_sum (tot[0] = total): device_scan.hpp
  const auto scan_u32 = [&](const uint32_t* a, uint64_t m, uint32_t* out_sum) { launch_scan_u32(s, a, m, sc_partial, tot, out_sum); };

  // --- bounding box (getMinMax3D) ---
  // (the per-block extremes are written straight into pinned host memory: the host waits for the kernel, not for a
  // copy command behind it)
  std::vector<float> hp(size_t(nb) * 6);
  {
    float* pinned = nullptr;
    const size_t bytes = hp.size() * sizeof(float);
    PCLHIP_CHECK_HIP(ctx, pinned_malloc(ctx, &pinned, bytes));
    hipLaunchKernelGGL(vg_minmax_kernel, dim3(nb), dim3(256), 0, s, dp, stride, n, has_z_limits, float(z_min), float(z_max),
                       pinned);
    const hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess) memcpy(hp.data(), pinned, bytes);
    pinned_free(ctx, pinned, bytes);
    PCLHIP_CHECK_HIP(ctx, e);
  }
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
  if (dims) {  // getMinBoxCoordinates / getMaxBoxCoordinates / getNrDivisions / getDivisionMultiplier (voxel_grid.h:326-344)
    for (int d = 0; d < 3; ++d) {
      dims->min_b[d] = g.min_b[d];
      dims->max_b[d] = g.min_b[d] + div_b[d] - 1;
      dims->div_b[d] = div_b[d];
      dims->divb_mul[d] = g.mul[d];
    }
  }
  if (dims_only) return PCLHIP_OK;
  const uint64_t ncells = uint64_t(div_b[0]) * uint64_t(div_b[1]) * uint64_t(div_b[2]);
  int32_t* d_layout = nullptr;
  if (leaf_layout) {
    PCLHIP_REQUIRE(ctx, leaf_layout_capacity >= ncells, "leaf_layout holds fewer than div_b[0] * div_b[1] * div_b[2] cells");
    d_layout = leaf_layout;
    if (!is_device_pointer(leaf_layout)) {
      PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &d_layout, ncells * sizeof(int32_t)));
      guard.p.push_back(d_layout);
    }
    PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(d_layout, 0xFF, ncells * sizeof(int32_t), s));   // -1: empty cell
  }
  const auto layout_out = [&]() -> pclhip_status {  // the host copy of the layout (every return path after this point)
    if (leaf_layout && d_layout != leaf_layout) {
      PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(leaf_layout, d_layout, ncells * sizeof(int32_t), hipMemcpyDeviceToHost, s));
      PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
    }
    return PCLHIP_OK;
  };

  // --- keys + stable sort.  Voxel ids are < div_b[0] * div_b[1] * div_b[2]: only that many key bits need sorting;
  // rejected points carry the all-ones key, kept apart by ONE extra bit above the ids.
  const int64_t nvox = int64_t(div_b[0]) * div_b[1] * div_b[2];
  int id_bits = 1;
  while (id_bits < 32 && (int64_t(1) << id_bits) < nvox) ++id_bits;
  const int sort_bits = id_bits < 32 ? id_bits + 1 : 32;
  PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(d_cnt, 0, sizeof(unsigned int), s));
  hipLaunchKernelGGL(vg_key_kernel, dim3(unsigned(std::min<uint64_t>((n + 255) / 256, uint64_t(ctx->num_cus) * 2))), dim3(256), 0,
                     s, dp, stride, n, g, has_z_limits, z_min, z_max, k0, d_cnt, sort_bits);
  for (int shift = 0; shift < sort_bits; shift += 8) {  // stable LSD passes (rs_* above), ping-pong k0/v0 <-> k1/v1
    hipLaunchKernelGGL(rs_hist_kernel, dim3(sort_blocks), dim3(RS_THREADS), 0, s, k0, uint32_t(n), shift, sort_blocks, hist);
    hipLaunchKernelGGL(rs_rowscan_kernel, dim3(256), dim3(256), 0, s, hist, sort_blocks, first, rowtot);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(sort_blocks), dim3(RS_THREADS), 0, s, k0,
                       shift == 0 ? static_cast<const uint32_t*>(nullptr) : v0, uint32_t(n), shift, sort_blocks, first, rowtot,
                       k1, v1);
    std::swap(k0, k1);
    std::swap(v0, v1);
  }
  const uint32_t* keys_sorted = k0;   // after the last swap
  const uint32_t* vals_sorted = v0;
  // --- runs --- (sized for all n points; the kernels read the number of valid ones from the device: one host
  // synchronisation for both counts instead of two)
  const uint32_t n32 = uint32_t(n);
  const uint32_t run_blocks = (n32 + uint32_t(VG_RUN_BLOCK) - 1u) / uint32_t(VG_RUN_BLOCK);
  uint32_t* block_heads = head;   // run_blocks counters and their exclusive scan: the front of two n-sized arrays
  uint32_t* block_first = scan;
  hipLaunchKernelGGL(vg_runcount_kernel, dim3(run_blocks), dim3(VG_RUN_THREADS), 0, s, keys_sorted, d_cnt, block_heads);
  scan_u32(block_heads, run_blocks, block_first);
  unsigned int nv = 0;
  uint32_t nruns = 0;
  hipLaunchKernelGGL(vg_runstart_kernel, dim3(run_blocks), dim3(VG_RUN_THREADS), 0, s, keys_sorted, d_cnt, block_first, run_start);
  {  // both counts for the host, through pinned memory (two copy commands in front of the kernel above cost the stream 45 us)
    uint32_t* pinned = nullptr;
    PCLHIP_CHECK_HIP(ctx, pinned_malloc(ctx, &pinned, 2 * sizeof(uint32_t)));
    hipLaunchKernelGGL(vg_counts_kernel, dim3(1), dim3(1), 0, s, d_cnt, tot, pinned);
    const hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess) {
      nv = pinned[0];
      nruns = pinned[1];
    }
    pinned_free(ctx, pinned, 2 * sizeof(uint32_t));
    PCLHIP_CHECK_HIP(ctx, e);
  }
  if (nv == 0) return layout_out();
  uint32_t* keep = head;        // head / scan are consumed: reuse them for the per-run arrays (nruns <= nv)
  uint32_t* keep_scan = scan;
  hipLaunchKernelGGL(vg_keep_kernel, dim3((nruns + 255) / 256), dim3(256), 0, s, run_start, nruns, min_points_per_voxel,
                     keep);
  scan_u32(keep, nruns, keep_scan);
  uint32_t total = nruns;
  if (min_points_per_voxel > 1) {  // otherwise every run is kept: no need to read the count back
    PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(&total, tot, sizeof total, hipMemcpyDeviceToHost, s));
    PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
  }

  if (d_layout) hipLaunchKernelGGL(vg_layout_kernel, dim3((nruns + 255) / 256), dim3(256), 0, s, keys_sorted, run_start, keep, keep_scan, nruns, d_layout);

  // --- centroids ---
  void* d_out = out;
  const bool out_dev = is_device_pointer(out);
  if (!out_dev) {
    PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &d_out, size_t(total > 0 ? total : 1) * out_stride));
    guard.p.push_back(d_out);
  }
  if (total > 0) {
    if (out_stride != 16 && (normals_offset == 0 || out_stride != normals_offset + 32))
      PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(d_out, 0, size_t(total) * out_stride, s));  // fields this filter does not fill
    if (uint64_t(nv) > uint64_t(nruns) * 8) {  // runs of a row's size and more: a row of 16 lanes per run
      const unsigned blocks = unsigned(std::min<uint64_t>((uint64_t(nruns) + 15) / 16, uint64_t(ctx->num_cus) * 16));
      hipLaunchKernelGGL(vg_centroid_row_kernel, dim3(blocks), dim3(256), 0, s, dp, stride, vals_sorted, run_start, keep,
                         keep_scan, nruns, d_out, out_stride, normals_offset, downsample_all_data ? 1 : 0);
    } else {
      hipLaunchKernelGGL(vg_centroid_kernel, dim3((nruns + 255) / 256), dim3(256), 0, s, dp, stride, vals_sorted, run_start, keep,
                         keep_scan, nruns, d_out, out_stride, normals_offset, downsample_all_data ? 1 : 0);
    }
    PCLHIP_CHECK_HIP(ctx, hipGetLastError());
    if (!out_dev)
      PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, d_out, size_t(total) * out_stride, hipMemcpyDeviceToHost, s));
  }
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
  *out_n = total;
  return layout_out();
}

extern "C" pclhip_status pclhip_voxelgrid_grid(pclhip_ctx* ctx, const void* points, size_t stride, uint64_t n, const float leaf[3],
                                               int has_z_limits, double z_min, double z_max, pclhip_voxelgrid_dims* dims) {
  if (!dims) return PCLHIP_ERR_INVALID;
  uint64_t none = 0;
  return pclhip_voxelgrid_ex2(ctx, points, stride, n, leaf, 0, has_z_limits, z_min, z_max, 1, 0, nullptr, 16, &none, nullptr, 0, dims);
}
