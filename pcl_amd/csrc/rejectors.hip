// rejectors.hip -- device-side stages between the correspondence search and the accumulation of an
// ICP iteration (registration/include/pcl/registration/impl/icp.hpp:176-201):
//   * reciprocal correspondences (impl/correspondence_estimation.hpp:220-311)
//   * the rejector chain: Distance, MedianDistance, OneToOne, Trimmed
//     (registration/src/correspondence_rejection_{distance,median_distance,one_to_one,trimmed}.cpp)
// All of them only clear bits of the per-slot `keep` mask; the matches themselves stay in place (they
// seed the next iteration).  Where the reference leaves the order of exact ties to an unstable
// std::sort, the lower query index wins.
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>

#include <cfloat>
#include <cmath>

#include "pclhip_internal.hpp"

namespace pclhip {
namespace {

constexpr int TB = 256;

// The per-pair flag passes take FOUR pairs per thread: one 16-byte load of match positions / distances and one 4-byte word of
// keep flags (a thread per pair moved one byte per lane and store: 19-20 us per pass at 10M pairs); thread n / 4 takes the
// n % 4 pairs left over.
__device__ __forceinline__ uint32_t rej_tail_first(uint32_t n) { return (n / 4u) * 4u; }

__global__ void rej_init_kernel(const uint32_t* __restrict__ match_pos, uint32_t n, uint8_t* __restrict__ keep) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, n4 = n / 4u;
  if (j < n4) {
    const uint4 m = reinterpret_cast<const uint4*>(match_pos)[j];
    reinterpret_cast<uint32_t*>(keep)[j] = (m.x != NO_INDEX ? 1u : 0u) | (m.y != NO_INDEX ? 0x100u : 0u) |
                                           (m.z != NO_INDEX ? 0x10000u : 0u) | (m.w != NO_INDEX ? 0x1000000u : 0u);
  } else if (j == n4) {
    for (uint32_t i = rej_tail_first(n); i < n; ++i) keep[i] = match_pos[i] != NO_INDEX ? 1 : 0;
  }
}

// registration/src/correspondence_rejection_distance.cpp:55-60: keep if distance < max_distance_^2 (float)
__global__ void rej_distance_kernel(const float* __restrict__ d2, uint32_t n, float max_d2, uint8_t* __restrict__ keep) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, n4 = n / 4u;
  if (j < n4) {
    const uint32_t k = reinterpret_cast<const uint32_t*>(keep)[j];
    if (k == 0u) return;
    const float4 d = reinterpret_cast<const float4*>(d2)[j];
    uint32_t o = k;
    if (!(d.x < max_d2)) o &= ~0xFFu;
    if (!(d.y < max_d2)) o &= ~0xFF00u;
    if (!(d.z < max_d2)) o &= ~0xFF0000u;
    if (!(d.w < max_d2)) o &= ~0xFF000000u;
    if (o != k) reinterpret_cast<uint32_t*>(keep)[j] = o;
  } else if (j == n4) {
    for (uint32_t i = rej_tail_first(n); i < n; ++i)
      if (keep[i] && !(d2[i] < max_d2)) keep[i] = 0;
  }
}

// ---- radix selection: ONE order statistic of the kept distances, read where they lie --------------------------
// MedianDistance and Trimmed need one order statistic each, not a sorted array.  The keys are never materialised:
// every digit pass reads the distances (4 B) and the keep bytes (1 B) -- 50 MB at 10M pairs -- builds a per-block LDS
// histogram of the kept distances that match the digits chosen so far and merges it into a global one; a one-workgroup
// kernel then picks the bin that holds the rank.  The 32 distance bits take three passes (10 + 11 + 11 bits; the bits
// of a non-negative float order like the float).  The FIRST pass also counts the kept pairs (the histogram's total), so
// the rank -- which depends on that count -- is derived inside the first pick; the LAST pick writes the threshold.
// Trimmed orders by (distance, original query index): when the distance holding the rank is shared by more pairs than
// the rank takes, three more passes select on the query index (read from the source record only where the distance
// equals the selected one); otherwise -- one pair at that distance, the usual case -- they return at once.
struct RsState {
  uint32_t dkey;  // distance bits decided so far (after the three distance passes: the distance itself)
  uint32_t ikey;  // query-index bits decided so far (Trimmed's tie passes)
  uint32_t rank;  // remaining rank inside the chosen bin
  uint32_t done;  // the selection is over (or there is nothing to select): later passes return at once
};
constexpr int RS_BINS = 2048;
constexpr int RS_PASSES = 6;
// selection passes (= histogram all-reduces of a sharded run) per rejector kind: the distance's three digits for the
// median, three more over the query index for Trimmed's ties.  ONE place: radix_select_queued runs them, a rank with an
// empty share issues as many zero histograms (apply_empty_shard_collectives).
constexpr int RS_PASSES_MEDIAN = 3, RS_PASSES_TRIMMED = 6;
constexpr int rs_passes_of(int kind) {
  return kind == PCLHIP_REJ_TRIMMED ? RS_PASSES_TRIMMED : kind == PCLHIP_REJ_MEDIAN_DISTANCE ? RS_PASSES_MEDIAN : 0;
}

template <int PASS>
__device__ __forceinline__ void rs_count(uint32_t* h, float d, bool kept, uint32_t dkey, uint32_t ikey, const float4* cur,
                                         uint32_t i) {
  if (!kept) return;
  const uint32_t b = __float_as_uint(d);
  if constexpr (PASS == 0) {
    atomicAdd(&h[b >> 22], 1u);
  } else if constexpr (PASS == 1) {
    if ((b >> 22) == (dkey >> 22)) atomicAdd(&h[(b >> 11) & 2047u], 1u);
  } else if constexpr (PASS == 2) {
    if ((b >> 11) == (dkey >> 11)) atomicAdd(&h[b & 2047u], 1u);
  } else {
    if (b != dkey) return;
    const uint32_t q = __float_as_uint(cur[i].w);
    if constexpr (PASS == 3) {
      atomicAdd(&h[q >> 22], 1u);
    } else if constexpr (PASS == 4) {
      if ((q >> 22) == (ikey >> 22)) atomicAdd(&h[(q >> 11) & 2047u], 1u);
    } else {
      if ((q >> 11) == (ikey >> 11)) atomicAdd(&h[q & 2047u], 1u);
    }
  }
}

template <int PASS>
__global__ __launch_bounds__(TB) void rs_hist_kernel(const float* __restrict__ d2, const uint8_t* __restrict__ keep,
                                                     const float4* __restrict__ cur, uint32_t n,
                                                     const RsState* __restrict__ st, uint32_t* __restrict__ hist) {
  uint32_t dkey = 0, ikey = 0;
  if constexpr (PASS > 0) {
    if (st->done) return;
    dkey = st->dkey;
    ikey = st->ikey;
  }
  __shared__ uint32_t h[RS_BINS];
  for (int i = threadIdx.x; i < RS_BINS; i += TB) h[i] = 0u;
  __syncthreads();
  const uint32_t n4 = n / 4;  // four pairs per lane and step: one 16-byte and one 4-byte load
  const float4* d4 = reinterpret_cast<const float4*>(d2);
  const uint32_t* k4 = reinterpret_cast<const uint32_t*>(keep);
  // RS_TRIPS trips of the grid-stride loop at a time: their flag words and distances are asked for together (the distance
  // load behind `if (k == 0) continue` was a second memory round trip per trip); slots past the end re-read the first
  constexpr int RS_TRIPS = 4;
  const uint32_t G = gridDim.x * TB;
  for (uint32_t j0 = blockIdx.x * TB + threadIdx.x; j0 < n4; j0 += RS_TRIPS * G) {
    uint32_t kk[RS_TRIPS];
    float4 dd[RS_TRIPS];
#pragma unroll
    for (int u = 0; u < RS_TRIPS; ++u) {
      const uint64_t j = uint64_t(j0) + uint64_t(u) * G;
      kk[u] = k4[j < n4 ? uint32_t(j) : j0];
    }
#pragma unroll
    for (int u = 0; u < RS_TRIPS; ++u) {
      const uint64_t j = uint64_t(j0) + uint64_t(u) * G;
      dd[u] = d4[j < n4 ? uint32_t(j) : j0];
    }
#pragma unroll
    for (int u = 0; u < RS_TRIPS; ++u) {
      const uint64_t j64 = uint64_t(j0) + uint64_t(u) * G;
      if (j64 >= n4) continue;
      const uint32_t j = uint32_t(j64), k = kk[u];
      if (k == 0u) continue;
      const float4 d = dd[u];
      rs_count<PASS>(h, d.x, (k & 0xFFu) != 0u, dkey, ikey, cur, 4 * j);
      rs_count<PASS>(h, d.y, (k & 0xFF00u) != 0u, dkey, ikey, cur, 4 * j + 1);
      rs_count<PASS>(h, d.z, (k & 0xFF0000u) != 0u, dkey, ikey, cur, 4 * j + 2);
      rs_count<PASS>(h, d.w, (k & 0xFF000000u) != 0u, dkey, ikey, cur, 4 * j + 3);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3u)) {
    const uint32_t i = 4 * n4 + threadIdx.x;
    rs_count<PASS>(h, d2[i], keep[i] != 0, dkey, ikey, cur, i);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < RS_BINS; i += TB)
    if (h[i]) atomicAdd(hist + i, h[i]);
}

// One workgroup: scan of the pass's histogram, the bin holding the rank, and what the pass's place in the chain asks for.
//   MedianDistance: rank = count / 2 (nth_element at size/2, correspondence_rejection_median_distance.cpp:55-56),
//                   threshold = median * factor in double (:64-66)
//   Trimmed:        nv = max(floor(float(ratio) * float(count)), min_correspondences) (correspondence_rejection_trimmed.cpp
//                   :47-50); nothing to do when nv >= count, everything dropped when nv == 0, else rank = nv - 1
// The count of kept pairs, the rank it implies, the selected key and the threshold derived from it stay in device memory
// (pclhip::RejState, one per registration; mirrored to pinned memory for the getters): the chain runs stream-ordered,
// without the host in between -- it is part of the device-driven loop.
template <int PASS>
__global__ __launch_bounds__(256) void rs_pick_kernel(RejState* __restrict__ out, RsState* __restrict__ st,
                                                      const uint32_t* __restrict__ hist, int kind, double param,
                                                      unsigned int min_corr) {
  if constexpr (PASS > 0) {
    if (st->done) return;  // (uniform over the workgroup; nobody has written st yet)
  }
  constexpr int PER = RS_BINS / 256;
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
  uint32_t rank;
  if constexpr (PASS == 0) {
    const unsigned int cnt = scan[255];  // every kept pair is in exactly one bin of the first pass
    int mode = 0;                        // 0 nothing to filter, 1 threshold at `rank`, 2 drop everything
    rank = 0;
    bool cut = false;
    if (kind == PCLHIP_REJ_MEDIAN_DISTANCE) {
      mode = cnt > 0 ? 1 : 0;
      rank = cnt / 2;
    } else {  // PCLHIP_REJ_TRIMMED
      const float prod = __fmul_rn(float(param), float(cnt));
      unsigned int nv = (unsigned int)floorf(prod);
      if (nv < min_corr) nv = min_corr;
      if (nv < cnt) {
        mode = nv == 0 ? 2 : 1;
        rank = nv == 0 ? 0 : nv - 1;
        cut = true;
      }
    }
    if (threadIdx.x == 0) {
      out->count = cnt;
      out->mode = mode;
      if (cut) out->trimmed = 1;
      st->dkey = 0;
      st->ikey = 0;
      st->rank = rank;
      st->done = mode == 1 ? 0u : 1u;
    }
    if (mode != 1) return;
  } else {
    rank = st->rank;
  }
  const uint32_t incl = scan[threadIdx.x], excl = incl - sum;
  __syncthreads();  // every thread has read st before one of them rewrites it
  if (rank >= excl && rank < incl) {
    uint32_t cum = excl;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      if (rank >= cum && rank < cum + c[j]) {
        const uint32_t bin = threadIdx.x * PER + j, left = rank - cum;
        constexpr int SHIFT = (PASS % 3 == 0) ? 22 : (PASS % 3 == 1) ? 11 : 0;
        if constexpr (PASS < 3) {
          const uint32_t dkey = (PASS == 0 ? 0u : st->dkey) | (bin << SHIFT);
          st->dkey = dkey;
          st->rank = left;
          if constexpr (PASS == 2) {
            if (kind == PCLHIP_REJ_MEDIAN_DISTANCE) {
              const double median = double(__uint_as_float(dkey));
              out->key = dkey;
              out->median = median;
              out->threshold = median * param;
              st->done = 1u;
            } else if (left + 1u == c[j]) {  // the rank takes every pair at this distance: no tie to order
              out->key = ((unsigned long long)dkey << 32) | 0xFFFFFFFFull;
              st->done = 1u;
            }
          }
        } else {
          const uint32_t ikey = (PASS == 3 ? 0u : st->ikey) | (bin << SHIFT);
          st->ikey = ikey;
          st->rank = left;
          if constexpr (PASS == 5) {
            out->key = ((unsigned long long)st->dkey << 32) | ikey;
            st->done = 1u;
          }
        }
      }
      cum += c[j];
    }
  }
}

// Multi-GPU: every rank histograms the pairs IT serves; the histograms are summed over the ranks between the histogram
// and the pick kernel (as doubles: the all-reduce of the record is the one collective the library has), so every rank
// picks the same bin from the same global counts -- the count of kept pairs, the rank, the threshold and the `done` flag
// are the single-GPU run's on every rank.  The host queues the same passes whatever `done` says, so the collectives match.
__global__ void rs_hist_to_f64_kernel(const uint32_t* __restrict__ h, double* __restrict__ d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < RS_BINS) d[i] = double(h[i]);
}
__global__ void rs_hist_from_f64_kernel(const double* __restrict__ d, uint32_t* __restrict__ h) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < RS_BINS) h[i] = uint32_t(d[i]);  // exact: counts of at most 2^31 pairs
}

// the order statistic a MedianDistance (three passes) or Trimmed (three + three tie passes) rejector asks for ->
// RejState (count, mode, key, median, threshold), stream-ordered.  hist_dev: RS_PASSES histograms.
template <int PASS>
pclhip_status radix_select_pass(pclhip_icp* icp, int grid, const float* d2, const uint8_t* keep, const float4* cur, uint32_t n,
                                RejState* out, RsState* rs, uint32_t* hist_dev, double* hist_f64, int kind, double param,
                                unsigned int min_corr) {
  pclhip_ctx* ctx = icp->ctx;
  uint32_t* const h = hist_dev + size_t(PASS) * RS_BINS;
  hipLaunchKernelGGL(rs_hist_kernel<PASS>, dim3(grid), dim3(TB), 0, ctx->stream, d2, keep, cur, n, rs, h);
  if (hist_f64 != nullptr) {
    hipLaunchKernelGGL(rs_hist_to_f64_kernel, dim3(RS_BINS / 256), dim3(256), 0, ctx->stream, h, hist_f64);
    const pclhip_status st = allreduce_doubles(icp, hist_f64, RS_BINS);
    if (st != PCLHIP_OK) return st;
    hipLaunchKernelGGL(rs_hist_from_f64_kernel, dim3(RS_BINS / 256), dim3(256), 0, ctx->stream, hist_f64, h);
  }
  hipLaunchKernelGGL(rs_pick_kernel<PASS>, dim3(1), dim3(256), 0, ctx->stream, out, rs, h, kind, param, min_corr);
  return PCLHIP_OK;
}
pclhip_status radix_select_queued(pclhip_icp* icp, const float* d2, const uint8_t* keep, const float4* cur, uint32_t n,
                                  RejState* out, RsState* rs, uint32_t* hist_dev, double* hist_f64, int kind, double param,
                                  unsigned int min_corr) {
  pclhip_ctx* ctx = icp->ctx;
  int grid = int((n / 4 + TB - 1) / TB);
  // (every block merges up to 2048 bins into the global histogram, and float distances crowd into a few of them: 8 / 4 / 2 /
  // 1 blocks per CU measured at 10M pairs -- 1.278 / 1.236 / 1.236 / 1.291 ms per step with a median + trimmed chain in
  // round 4; with four trips of loads in flight per thread (round 6) four per CU 1.122-1.133, two per CU 1.104)
  if (grid > ctx->num_cus * 2) grid = ctx->num_cus * 2;
  if (grid < 1) grid = 1;
  PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(hist_dev, 0, size_t(RS_PASSES) * RS_BINS * sizeof(uint32_t), ctx->stream));
#define RS_PASS(P)                                                                                                    \
  do {                                                                                                                \
    const pclhip_status st_ = radix_select_pass<P>(icp, grid, d2, keep, cur, n, out, rs, hist_dev, hist_f64, kind, param, \
                                                   min_corr);                                                         \
    if (st_ != PCLHIP_OK) return st_;                                                                                 \
  } while (0)
  // (the number of passes -- and with it the number of collectives of a sharded run -- is rs_passes_of(kind): the ranks
  // with an empty share issue exactly that many, apply_empty_shard_collectives)
  RS_PASS(0);
  RS_PASS(1);
  RS_PASS(2);
  static_assert(RS_PASSES_MEDIAN == 3 && RS_PASSES_TRIMMED == 6 && RS_PASSES_TRIMMED <= RS_PASSES, "pass schedule");
  if (rs_passes_of(kind) == RS_PASSES_TRIMMED) {
    RS_PASS(3);
    RS_PASS(4);
    RS_PASS(5);
  }
#undef RS_PASS
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  return PCLHIP_OK;
}

// correspondence_rejection_median_distance.cpp:64-66: keep if double(d) <= median * factor
__global__ void rej_median_kernel(const float* __restrict__ d2, uint32_t n, const RejState* __restrict__ st,
                                  uint8_t* __restrict__ keep) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, n4 = n / 4u;
  if (st->mode != 1) return;
  const double thr = st->threshold;
  if (j < n4) {   // (four pairs per thread: see rej_init_kernel)
    const uint32_t k = reinterpret_cast<const uint32_t*>(keep)[j];
    if (k == 0u) return;
    const float4 d = reinterpret_cast<const float4*>(d2)[j];
    uint32_t o = k;
    if (!(double(d.x) <= thr)) o &= ~0xFFu;
    if (!(double(d.y) <= thr)) o &= ~0xFF00u;
    if (!(double(d.z) <= thr)) o &= ~0xFF0000u;
    if (!(double(d.w) <= thr)) o &= ~0xFF000000u;
    if (o != k) reinterpret_cast<uint32_t*>(keep)[j] = o;
  } else if (j == n4) {
    for (uint32_t i = rej_tail_first(n); i < n; ++i)
      if (keep[i] && !(double(d2[i]) <= thr)) keep[i] = 0;
  }
}

// correspondence_rejection_trimmed.cpp:53-58: keep the n smallest (distance, query) keys
__device__ __forceinline__ bool rej_trim_drops(int mode, uint32_t thr_d, uint32_t thr_q, float d, const float4* __restrict__ cur,
                                               uint32_t i) {
  const uint32_t b = __float_as_uint(d);  // the query index is read only where the distance alone does not decide
  return mode == 2 || b > thr_d || (b == thr_d && thr_q != 0xFFFFFFFFu && __float_as_uint(cur[i].w) > thr_q);
}
__global__ void rej_trim_kernel(const float4* __restrict__ cur, const float* __restrict__ d2, uint32_t n,
                                const RejState* __restrict__ st, uint8_t* __restrict__ keep) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, n4 = n / 4u;
  const int mode = st->mode;
  if (mode == 0) return;  // nv >= count: the list passes unchanged
  const uint64_t thr_key = st->key;
  const uint32_t thr_d = uint32_t(thr_key >> 32), thr_q = uint32_t(thr_key);
  if (j < n4) {   // (four pairs per thread: see rej_init_kernel)
    const uint32_t k = reinterpret_cast<const uint32_t*>(keep)[j];
    if (k == 0u) return;
    const float4 d = reinterpret_cast<const float4*>(d2)[j];
    uint32_t o = k;
    if ((k & 0xFFu) && rej_trim_drops(mode, thr_d, thr_q, d.x, cur, 4u * j)) o &= ~0xFFu;
    if ((k & 0xFF00u) && rej_trim_drops(mode, thr_d, thr_q, d.y, cur, 4u * j + 1u)) o &= ~0xFF00u;
    if ((k & 0xFF0000u) && rej_trim_drops(mode, thr_d, thr_q, d.z, cur, 4u * j + 2u)) o &= ~0xFF0000u;
    if ((k & 0xFF000000u) && rej_trim_drops(mode, thr_d, thr_q, d.w, cur, 4u * j + 3u)) o &= ~0xFF000000u;
    if (o != k) reinterpret_cast<uint32_t*>(keep)[j] = o;
  } else if (j == n4) {
    for (uint32_t i = rej_tail_first(n); i < n; ++i)
      if (keep[i] && rej_trim_drops(mode, thr_d, thr_q, d2[i], cur, i)) keep[i] = 0;
  }
}

// correspondence_rejection_one_to_one.cpp:49-65: per match index the smallest (distance, query) wins
__global__ void rej_o2o_min_kernel(const float4* __restrict__ cur, const uint32_t* __restrict__ match,
                                   const float* __restrict__ d2, const uint8_t* __restrict__ keep, uint32_t n,
                                   unsigned long long* __restrict__ best) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && keep[i]) {
    const unsigned long long k = (uint64_t(__float_as_uint(d2[i])) << 32) | __float_as_uint(cur[i].w);
    atomicMin(best + match[i], k);
  }
}
__global__ void rej_o2o_keep_kernel(const float4* __restrict__ cur, const uint32_t* __restrict__ match,
                                    const float* __restrict__ d2, uint32_t n,
                                    const unsigned long long* __restrict__ best, uint8_t* __restrict__ keep) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && keep[i]) {
    const unsigned long long k = (uint64_t(__float_as_uint(d2[i])) << 32) | __float_as_uint(cur[i].w);
    if (best[match[i]] != k) keep[i] = 0;
  }
}


struct Guard {
  pclhip_ctx* ctx = nullptr;
  std::vector<void*> p;
  ~Guard() {
    for (void* q : p)
      if (q) (void)dev_free(ctx, q);
  }
  template <class T>
  hipError_t alloc(T** ptr, size_t bytes) {
    hipError_t e = dev_malloc(ctx, ptr, bytes ? bytes : 16);
    if (e == hipSuccess) p.push_back(*ptr);
    return e;
  }
};

}  // namespace

// Nothing here waits for the stream (the chain is part of the device-driven loop); the one exception is the first use
// of reciprocal correspondences with a source cloud, which builds the source index.
pclhip_status apply_correspondence_filters(pclhip_icp* icp, float max_d2, bool use_max) {
  pclhip_ctx* ctx = icp->ctx;
  hipStream_t s = ctx->stream;
  const uint32_t n = icp->n;
  icp->fetch_order = 0;
  if (n == 0) return PCLHIP_OK;
  const dim3 grid((n + TB - 1) / TB), block(TB);
  const dim3 grid4((n / 4 + 1 + TB - 1) / TB);   // the per-pair flag passes: four pairs per thread + one thread for the tail
  if (!icp->keep) PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &icp->keep, n));
  if (!icp->rej_state) {
    PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &icp->rej_state, sizeof(RejState)));
    PCLHIP_CHECK_HIP(ctx, pinned_malloc(ctx, &icp->rej_state_host, sizeof(RejState)));
    std::memset(icp->rej_state_host, 0, sizeof(RejState));
  }
  PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(icp->rej_state, 0, sizeof(RejState), s));
  hipLaunchKernelGGL(rej_init_kernel, grid4, block, 0, s, icp->match_pos, n, icp->keep);
  Guard g;  // stream-ordered temporaries: released to the context when this returns, re-used only by later work of the stream
  g.ctx = ctx;
  RsState* rs = nullptr;
  uint32_t* rs_hist = nullptr;
  PCLHIP_CHECK_HIP(ctx, g.alloc(&rs, sizeof(RsState)));
  PCLHIP_CHECK_HIP(ctx, g.alloc(&rs_hist, size_t(RS_PASSES) * RS_BINS * sizeof(uint32_t)));
  double* rs_hist_f64 = nullptr;  // multi-GPU: the histogram of a pass as doubles, summed over the ranks
  if (icp_is_sharded(icp)) PCLHIP_CHECK_HIP(ctx, g.alloc(&rs_hist_f64, RS_BINS * sizeof(double)));

  if (icp->reciprocal) {
    // An index over the source, in the order the source already has: the registration's kd-ordered working copy IS the
    // index's point array (build_index_over: nothing sorted, nothing copied), and every iteration its leaf blocks and
    // boxes are recomputed from the moved cloud (refit_boxes: one pass over the cloud), stream-ordered.  The answers are
    // exact nearest neighbours with (distance, index) ties like any other index's: they do not depend on how well the
    // boxes fit.  (Until round 4 the index had its own copy of the points in its own order, gathered from the working
    // copy every iteration through two index arrays: 0.25 ms of random access per iteration at 10M points.)
    if (icp->src_index == nullptr) {
      pclhip_status st = build_index_over(ctx, icp->src_cur, icp->n_finite, uint32_t(icp->n_orig), &icp->src_index);  // synchronises: once
      if (st != PCLHIP_OK) return st;
    } else if (icp->src_index->n > 0) {
      pclhip_status st = refit_boxes(icp->src_index);
      if (st != PCLHIP_OK) return st;
    }
    // one seeded search of the source index per surviving pair; the test itself is fused into it (search.hip); the seed of
    // slot i is the source point i itself (the index's positions are the slots)
    const pclhip_status st = launch_recip_search(icp->src_index, icp->target->pts, icp->match_pos, icp->src_cur, n, max_d2,
                                                 use_max, icp->keep);
    if (st != PCLHIP_OK) return st;
  }

  bool trimmed_in_chain = false;
  for (const pclhip_rejector& r : icp->rejectors) {
    switch (r.kind) {
      case PCLHIP_REJ_DISTANCE: {
        const float md = float(r.param);
        hipLaunchKernelGGL(rej_distance_kernel, grid4, block, 0, s, icp->match_d2, n, md * md, icp->keep);
        break;
      }
      case PCLHIP_REJ_MEDIAN_DISTANCE: {
        {
          const pclhip_status sr = radix_select_queued(icp, icp->match_d2, icp->keep, icp->src_cur, n, icp->rej_state, rs,
                                                       rs_hist, rs_hist_f64, int(r.kind), r.param, 0u);
          if (sr != PCLHIP_OK) return sr;
        }
        hipLaunchKernelGGL(rej_median_kernel, grid4, block, 0, s, icp->match_d2, n, icp->rej_state, icp->keep);
        break;
      }
      case PCLHIP_REJ_ONE_TO_ONE: {
        unsigned long long* best = nullptr;
        const size_t nt = size_t(icp->target->n_orig);
        PCLHIP_CHECK_HIP(ctx, g.alloc(&best, nt * 8));
        PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(best, 0xFF, nt * 8, s));
        hipLaunchKernelGGL(rej_o2o_min_kernel, grid, block, 0, s, icp->src_cur, icp->match, icp->match_d2, icp->keep, n,
                           best);
        if (icp_is_sharded(icp)) {  // target slabs: a target point of two halos may be matched on two ranks -- the global minimum wins
          // best[] is indexed by the GLOBAL target index: every rank must have built its index over the same whole cloud
          // plus its own subset list (pclhip.h, pclhip_icp_set_region) -- checked once per registration, not assumed
          const pclhip_status sc = check_same_target_size(icp);
          if (sc != PCLHIP_OK) return sc;
          const pclhip_status sm = allreduce_min_u64(icp, best, nt);
          if (sm != PCLHIP_OK) return sm;
        }
        hipLaunchKernelGGL(rej_o2o_keep_kernel, grid, block, 0, s, icp->src_cur, icp->match, icp->match_d2, n, best,
                           icp->keep);
        icp->fetch_order = 1;
        trimmed_in_chain = false;  // whatever an earlier Trimmed did to the order, this one re-orders the list
        break;
      }
      case PCLHIP_REJ_TRIMMED: {
        // the nv-th smallest (distance, query) key
        {
          const pclhip_status sr = radix_select_queued(icp, icp->match_d2, icp->keep, icp->src_cur, n, icp->rej_state, rs,
                                                       rs_hist, rs_hist_f64, int(r.kind), r.param, r.min_correspondences);
          if (sr != PCLHIP_OK) return sr;
        }
        hipLaunchKernelGGL(rej_trim_kernel, grid4, block, 0, s, icp->src_cur, icp->match_d2, n, icp->rej_state, icp->keep);
        trimmed_in_chain = true;  // the list comes back sorted by distance IF it was cut (RejState::trimmed says so)
        break;
      }
      default:
        set_error(ctx, "unknown rejector kind");
        return PCLHIP_ERR_INVALID;
    }
  }
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  icp->trim_pending = trimmed_in_chain;
  // the getters (last median, order of the fetched list) read the pinned mirror after a synchronisation of their own
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(icp->rej_state_host, icp->rej_state, sizeof(RejState), hipMemcpyDeviceToHost, s));
  return PCLHIP_OK;
}

// A rank of a sharded registration whose source share is EMPTY (fewer source points than ranks: dist.shard_range gives
// such ranks a count of 0) has nothing to filter, but its peers all-reduce one 2048-bin histogram per selection pass of
// every MedianDistance (3 passes) and Trimmed (6) rejector between their histogram and pick kernels: the empty rank issues
// the same collectives, in the same order, with zero histograms -- otherwise RCCL sees mismatched operations and hangs
// (ADVICE r4, medium).  Stream-ordered; no-op for a single-GPU registration.
pclhip_status apply_empty_shard_collectives(pclhip_icp* icp) {
  if (!icp_is_sharded(icp)) return PCLHIP_OK;
  pclhip_ctx* ctx = icp->ctx;
  int passes = 0;
  for (const pclhip_rejector& r : icp->rejectors) {
    passes += rs_passes_of(r.kind);
    // the only other collective of the chain is OneToOne's minimum, which sharded_filters_ok admits with a sharded TARGET
    // only -- where every rank holds the whole source, so no rank's share is empty.  Were that ever relaxed, this rank
    // would have to issue that collective too: refuse instead of hanging RCCL.
    if (r.kind == PCLHIP_REJ_ONE_TO_ONE) {
      set_error(ctx, "OneToOne on a rank with an empty source share: its collective is not issued here");
      return PCLHIP_ERR_STATE;
    }
  }
  if (passes == 0) return PCLHIP_OK;
  Guard g;
  g.ctx = ctx;
  double* zeros = nullptr;
  PCLHIP_CHECK_HIP(ctx, g.alloc(&zeros, RS_BINS * sizeof(double)));
  for (int p = 0; p < passes; ++p) {
    PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(zeros, 0, RS_BINS * sizeof(double), ctx->stream));
    const pclhip_status st = allreduce_doubles(icp, zeros, RS_BINS);
    if (st != PCLHIP_OK) return st;
  }
  return PCLHIP_OK;
}

void preload_rejector_kernels() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(rs_pick_kernel<0>));
}
}  // namespace pclhip
