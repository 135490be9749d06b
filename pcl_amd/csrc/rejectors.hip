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

__global__ void rej_init_kernel(const uint32_t* __restrict__ match_pos, uint32_t n, uint8_t* __restrict__ keep) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keep[i] = match_pos[i] != NO_INDEX ? 1 : 0;
}

// registration/src/correspondence_rejection_distance.cpp:55-60: keep if distance < max_distance_^2 (float)
__global__ void rej_distance_kernel(const float* __restrict__ d2, uint32_t n, float max_d2, uint8_t* __restrict__ keep) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && keep[i] && !(d2[i] < max_d2)) keep[i] = 0;
}

// order-preserving key of a non-negative float, +inf (0xFFFFFFFF) for dropped slots
__global__ void rej_dist_key_kernel(const float* __restrict__ d2, const uint8_t* __restrict__ keep, uint32_t n,
                                    uint32_t* __restrict__ keys, unsigned int* __restrict__ count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool k = false;
  if (i < n) {
    k = keep[i] != 0;
    keys[i] = k ? __float_as_uint(d2[i]) : 0xFFFFFFFFu;
  }
  __shared__ unsigned int blk;
  if (threadIdx.x == 0) blk = 0;
  __syncthreads();
  const unsigned long long b = __builtin_amdgcn_ballot_w64(k);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(&blk, (unsigned int)__builtin_popcountll(b));
  __syncthreads();
  if (threadIdx.x == 0 && blk) atomicAdd(count, blk);
}

// ---- radix selection: the key of rank r among n unsigned keys ------------------------------------------------
// MedianDistance and Trimmed only need ONE order statistic of the kept distances, not a sorted array: digit
// passes of 11 bits from the top, each a histogram of the keys that match the prefix chosen so far (per-block
// LDS histogram merged into a global one) followed by a one-workgroup scan that picks the bin holding the rank.
struct RsState {
  unsigned long long prefix;  // key bits decided so far (after the last pass: the key itself)
  uint32_t rank;              // remaining rank inside the chosen bin
  uint32_t pad;
};
constexpr int RS_BINS = 2048;

template <class K>
__global__ __launch_bounds__(TB) void rs_hist_kernel(const K* __restrict__ keys, uint32_t n, const RsState* __restrict__ st,
                                                     int shift, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[RS_BINS];
  for (int i = threadIdx.x; i < RS_BINS; i += TB) h[i] = 0u;
  __syncthreads();
  constexpr int BITS = int(sizeof(K)) * 8;
  const int up = shift + 11;  // the bits above the digit must equal the prefix
  const unsigned long long prefix = st->prefix;
  for (uint32_t i = blockIdx.x * TB + threadIdx.x; i < n; i += gridDim.x * TB) {
    const unsigned long long k = keys[i];
    if (up >= BITS || (k >> up) == (prefix >> up)) atomicAdd(&h[uint32_t(k >> shift) & uint32_t(RS_BINS - 1)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < RS_BINS; i += TB)
    if (h[i]) atomicAdd(hist + i, h[i]);
}

__global__ __launch_bounds__(256) void rs_pick_kernel(RsState* __restrict__ st, const uint32_t* __restrict__ hist, int shift) {
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
  const uint32_t rank = st->rank;
  const uint32_t incl = scan[threadIdx.x], excl = incl - sum;
  __syncthreads();  // every thread has read st->rank before one of them rewrites it
  if (rank >= excl && rank < incl) {
    uint32_t cum = excl;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      if (rank >= cum && rank < cum + c[j]) {
        st->prefix |= (unsigned long long)(threadIdx.x * PER + j) << shift;
        st->rank = rank - cum;
      }
      cum += c[j];
    }
  }
}

// The chain runs stream-ordered, without the host in between (it is part of the device-driven loop): the count of kept
// pairs, the rank it implies, the selected key and the threshold derived from it stay in device memory (pclhip::RejState,
// one per registration; mirrored to pinned memory for the getters).
//   MedianDistance: rank = count / 2 (nth_element at size/2, correspondence_rejection_median_distance.cpp:55-56),
//                   threshold = median * factor in double (:64-66)
//   Trimmed:        nv = max(floor(float(ratio) * float(count)), min_correspondences) (correspondence_rejection_trimmed.cpp
//                   :47-50); nothing to do when nv >= count, everything dropped when nv == 0, else rank = nv - 1
__global__ void rej_prepare_kernel(RejState* __restrict__ st, RsState* __restrict__ rs, const unsigned int* __restrict__ count,
                                   int kind, double param, unsigned int min_corr) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const unsigned int cnt = *count;
  unsigned int rank = 0;
  int mode = 0;  // 0 nothing to filter, 1 threshold at `rank`, 2 drop everything
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
      st->trimmed = 1;
    }
  }
  st->count = cnt;
  st->mode = mode;
  rs->prefix = 0;
  rs->rank = rank;
  rs->pad = 0;
}

// after the digit passes: the selected key, and what MedianDistance makes of it
__global__ void rej_threshold_kernel(RejState* __restrict__ st, const RsState* __restrict__ rs, int kind, double param) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  st->key = rs->prefix;
  if (kind == PCLHIP_REJ_MEDIAN_DISTANCE && st->mode == 1) {
    const double median = double(__uint_as_float(uint32_t(rs->prefix)));
    st->median = median;
    st->threshold = median * param;
  }
}

// key of the rank rs->rank (set by rej_prepare_kernel) -> rs->prefix, stream-ordered
template <class K>
void radix_select_queued(pclhip_ctx* ctx, const K* keys, uint32_t n, RsState* rs, uint32_t* hist_dev) {
  hipStream_t s = ctx->stream;
  constexpr int BITS = int(sizeof(K)) * 8;
  int grid = int((n + TB - 1) / TB);
  if (grid > ctx->num_cus * 8) grid = ctx->num_cus * 8;
  for (int shift = ((BITS - 1) / 11) * 11; shift >= 0; shift -= 11) {
    (void)hipMemsetAsync(hist_dev, 0, RS_BINS * sizeof(uint32_t), s);
    hipLaunchKernelGGL(rs_hist_kernel<K>, dim3(grid), dim3(TB), 0, s, keys, n, rs, shift, hist_dev);
    hipLaunchKernelGGL(rs_pick_kernel, dim3(1), dim3(256), 0, s, rs, hist_dev, shift);
  }
}

// correspondence_rejection_median_distance.cpp:64-66: keep if double(d) <= median * factor
__global__ void rej_median_kernel(const float* __restrict__ d2, uint32_t n, const RejState* __restrict__ st,
                                  uint8_t* __restrict__ keep) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (st->mode != 1) return;
  const double thr = st->threshold;
  if (i < n && keep[i] && !(double(d2[i]) <= thr)) keep[i] = 0;
}

// (distance, original query index) key; ~0 for dropped slots
__global__ void rej_pair_key_kernel(const float4* __restrict__ cur, const float* __restrict__ d2,
                                    const uint8_t* __restrict__ keep, uint32_t n, uint64_t* __restrict__ keys,
                                    unsigned int* __restrict__ count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool k = false;
  if (i < n) {
    k = keep[i] != 0;
    keys[i] = k ? ((uint64_t(__float_as_uint(d2[i])) << 32) | __float_as_uint(cur[i].w)) : ~0ull;
  }
  __shared__ unsigned int blk;
  if (threadIdx.x == 0) blk = 0;
  __syncthreads();
  const unsigned long long b = __builtin_amdgcn_ballot_w64(k);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(&blk, (unsigned int)__builtin_popcountll(b));
  __syncthreads();
  if (threadIdx.x == 0 && blk) atomicAdd(count, blk);
}

// correspondence_rejection_trimmed.cpp:53-58: keep the n smallest (distance, query) keys
__global__ void rej_trim_kernel(const float4* __restrict__ cur, const float* __restrict__ d2, uint32_t n,
                                const RejState* __restrict__ st, uint8_t* __restrict__ keep) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const int mode = st->mode;
  if (mode == 0) return;  // nv >= count: the list passes unchanged
  const uint64_t thr_key = st->key;
  if (i < n && keep[i]) {
    const uint64_t k = (uint64_t(__float_as_uint(d2[i])) << 32) | __float_as_uint(cur[i].w);
    if (mode == 2 || k > thr_key) keep[i] = 0;
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

// slot of every original source index inside the kd-ordered source arrays (src_sorted0 / src_cur)
__global__ void recip_slot_kernel(const float4* __restrict__ src_sorted0, uint32_t n, uint32_t* __restrict__ slot_of_orig) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) slot_of_orig[__float_as_uint(src_sorted0[j].w)] = j;
}
// the source index's points (its own kd order, w = original index) take the coordinates the cloud has NOW
__global__ void recip_gather_kernel(float4* __restrict__ ix_pts, uint32_t n, const uint32_t* __restrict__ slot_of_orig,
                                    const float4* __restrict__ cur) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float w = ix_pts[i].w;
    const float4 c = cur[slot_of_orig[__float_as_uint(w)]];
    ix_pts[i] = make_float4(c.x, c.y, c.z, w);
  }
}
// position of every slot's point inside the source index (w = original index on both sides)
__global__ void recip_pos_kernel(const float4* __restrict__ ix_pts, uint32_t n, const uint32_t* __restrict__ slot_of_orig,
                                 uint32_t* __restrict__ pos_of_slot) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) pos_of_slot[slot_of_orig[__float_as_uint(ix_pts[p].w)]] = p;
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
  if (!icp->keep) PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &icp->keep, n));
  if (!icp->rej_state) {
    PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &icp->rej_state, sizeof(RejState)));
    PCLHIP_CHECK_HIP(ctx, pinned_malloc(ctx, &icp->rej_state_host, sizeof(RejState)));
    std::memset(icp->rej_state_host, 0, sizeof(RejState));
  }
  PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(icp->rej_state, 0, sizeof(RejState), s));
  hipLaunchKernelGGL(rej_init_kernel, grid, block, 0, s, icp->match_pos, n, icp->keep);
  Guard g;  // stream-ordered temporaries: released to the context when this returns, re-used only by later work of the stream
  g.ctx = ctx;
  unsigned int* d_cnt = nullptr;
  RsState* rs = nullptr;
  uint32_t* rs_hist = nullptr;
  PCLHIP_CHECK_HIP(ctx, g.alloc(&d_cnt, sizeof(unsigned int)));
  PCLHIP_CHECK_HIP(ctx, g.alloc(&rs, sizeof(RsState)));
  PCLHIP_CHECK_HIP(ctx, g.alloc(&rs_hist, RS_BINS * sizeof(uint32_t)));

  if (icp->reciprocal) {
    // An index over the source: built ONCE per source cloud (from the pristine, kd-ordered copy; ids = original source
    // indices), then refitted to the moved cloud every iteration -- its points are gathered from src_cur and only the
    // boxes are recomputed (refit_boxes), stream-ordered.  The answers are exact nearest neighbours with (distance,
    // index) ties like any other index's: they do not depend on how well the boxes fit.
    if (icp->src_index == nullptr) {
      pclhip_status st = build_index_from_float4(ctx, icp->src_sorted0, n, &icp->src_index);  // synchronises: once
      if (st != PCLHIP_OK) return st;
      if (icp->src_index->disc) (void)dev_free(ctx, icp->src_index->disc);  // discs describe the cloud where it was built
      icp->src_index->disc = nullptr;
      PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &icp->src_slot_of_orig, size_t(icp->n_orig > 0 ? icp->n_orig : 1) * 4));
      hipLaunchKernelGGL(recip_slot_kernel, grid, block, 0, s, icp->src_sorted0, n, icp->src_slot_of_orig);
      // where the index keeps every slot's point: the seed of the slot's reciprocal search (points the index left out --
      // non-finite ones -- have no match and are never asked for)
      PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &icp->src_pos_of_slot, size_t(n) * 4));
      PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(icp->src_pos_of_slot, 0xFF, size_t(n) * 4, s));
      if (icp->src_index->n > 0)
        hipLaunchKernelGGL(recip_pos_kernel, dim3((icp->src_index->n + TB - 1) / TB), block, 0, s, icp->src_index->pts,
                           icp->src_index->n, icp->src_slot_of_orig, icp->src_pos_of_slot);
    }
    pclhip_index* const src_ix = icp->src_index;
    if (src_ix->n > 0) {
      hipLaunchKernelGGL(recip_gather_kernel, dim3((src_ix->n + TB - 1) / TB), block, 0, s, src_ix->pts, src_ix->n,
                         icp->src_slot_of_orig, icp->src_cur);
      pclhip_status st = refit_boxes(src_ix);
      if (st != PCLHIP_OK) return st;
    }
    // one seeded search of the source index per surviving pair; the test itself is fused into it (search.hip)
    const pclhip_status st = launch_recip_search(src_ix, icp->target->pts, icp->match_pos, icp->src_pos_of_slot, icp->src_cur,
                                                 n, max_d2, use_max, icp->keep);
    if (st != PCLHIP_OK) return st;
  }

  bool trimmed_in_chain = false;
  for (const pclhip_rejector& r : icp->rejectors) {
    switch (r.kind) {
      case PCLHIP_REJ_DISTANCE: {
        const float md = float(r.param);
        hipLaunchKernelGGL(rej_distance_kernel, grid, block, 0, s, icp->match_d2, n, md * md, icp->keep);
        break;
      }
      case PCLHIP_REJ_MEDIAN_DISTANCE: {
        uint32_t* k0 = nullptr;
        PCLHIP_CHECK_HIP(ctx, g.alloc(&k0, size_t(n) * 4));
        PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(d_cnt, 0, sizeof(unsigned int), s));
        // dropped slots carry the largest key, so rank count / 2 counts kept distances only
        hipLaunchKernelGGL(rej_dist_key_kernel, grid, block, 0, s, icp->match_d2, icp->keep, n, k0, d_cnt);
        hipLaunchKernelGGL(rej_prepare_kernel, dim3(1), dim3(1), 0, s, icp->rej_state, rs, d_cnt, int(r.kind), r.param, 0u);
        radix_select_queued<uint32_t>(ctx, k0, n, rs, rs_hist);
        hipLaunchKernelGGL(rej_threshold_kernel, dim3(1), dim3(1), 0, s, icp->rej_state, rs, int(r.kind), r.param);
        hipLaunchKernelGGL(rej_median_kernel, grid, block, 0, s, icp->match_d2, n, icp->rej_state, icp->keep);
        break;
      }
      case PCLHIP_REJ_ONE_TO_ONE: {
        unsigned long long* best = nullptr;
        const size_t nt = size_t(icp->target->n_orig);
        PCLHIP_CHECK_HIP(ctx, g.alloc(&best, nt * 8));
        PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(best, 0xFF, nt * 8, s));
        hipLaunchKernelGGL(rej_o2o_min_kernel, grid, block, 0, s, icp->src_cur, icp->match, icp->match_d2, icp->keep, n,
                           best);
        hipLaunchKernelGGL(rej_o2o_keep_kernel, grid, block, 0, s, icp->src_cur, icp->match, icp->match_d2, n, best,
                           icp->keep);
        icp->fetch_order = 1;
        trimmed_in_chain = false;  // whatever an earlier Trimmed did to the order, this one re-orders the list
        break;
      }
      case PCLHIP_REJ_TRIMMED: {
        uint64_t* k0 = nullptr;
        PCLHIP_CHECK_HIP(ctx, g.alloc(&k0, size_t(n) * 8));
        PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(d_cnt, 0, sizeof(unsigned int), s));
        hipLaunchKernelGGL(rej_pair_key_kernel, grid, block, 0, s, icp->src_cur, icp->match_d2, icp->keep, n, k0, d_cnt);
        hipLaunchKernelGGL(rej_prepare_kernel, dim3(1), dim3(1), 0, s, icp->rej_state, rs, d_cnt, int(r.kind), r.param,
                           r.min_correspondences);
        radix_select_queued<uint64_t>(ctx, k0, n, rs, rs_hist);  // the nv-th smallest (distance, query) key
        hipLaunchKernelGGL(rej_threshold_kernel, dim3(1), dim3(1), 0, s, icp->rej_state, rs, int(r.kind), r.param);
        hipLaunchKernelGGL(rej_trim_kernel, grid, block, 0, s, icp->src_cur, icp->match_d2, n, icp->rej_state, icp->keep);
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

void preload_rejector_kernels() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(rs_pick_kernel));
}
}  // namespace pclhip
