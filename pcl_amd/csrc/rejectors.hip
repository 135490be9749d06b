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

// key of rank `rank` (0-based, rank < n) -> *out; synchronises the stream
template <class K>
pclhip_status radix_select(pclhip_ctx* ctx, const K* keys, uint32_t n, uint32_t rank, RsState* st_dev, uint32_t* hist_dev,
                           K* out) {
  hipStream_t s = ctx->stream;
  RsState h;
  h.prefix = 0;
  h.rank = rank;
  h.pad = 0;
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(st_dev, &h, sizeof h, hipMemcpyHostToDevice, s));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));  // `h` is a stack object
  constexpr int BITS = int(sizeof(K)) * 8;
  int grid = int((n + TB - 1) / TB);
  if (grid > ctx->num_cus * 8) grid = ctx->num_cus * 8;
  for (int shift = ((BITS - 1) / 11) * 11; shift >= 0; shift -= 11) {
    PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(hist_dev, 0, RS_BINS * sizeof(uint32_t), s));
    hipLaunchKernelGGL(rs_hist_kernel<K>, dim3(grid), dim3(TB), 0, s, keys, n, st_dev, shift, hist_dev);
    hipLaunchKernelGGL(rs_pick_kernel, dim3(1), dim3(256), 0, s, st_dev, hist_dev, shift);
  }
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(&h, st_dev, sizeof h, hipMemcpyDeviceToHost, s));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
  *out = K(h.prefix);
  return PCLHIP_OK;
}

// correspondence_rejection_median_distance.cpp:64-66: keep if double(d) <= median * factor
__global__ void rej_median_kernel(const float* __restrict__ d2, uint32_t n, double thr, uint8_t* __restrict__ keep) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
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
                                uint64_t thr_key, uint8_t* __restrict__ keep) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && keep[i]) {
    const uint64_t k = (uint64_t(__float_as_uint(d2[i])) << 32) | __float_as_uint(cur[i].w);
    if (k > thr_key) keep[i] = 0;
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

// reciprocal: queries = matched target points, id = this slot's original source index
__global__ void recip_query_kernel(const float4* __restrict__ tgt_pts, const uint32_t* __restrict__ match_pos,
                                   const uint8_t* __restrict__ keep, uint32_t n, float4* __restrict__ q) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float qn = __builtin_nanf("");
  float4 v = make_float4(qn, qn, qn, __uint_as_float(i));  // .w = output row (this slot)
  if (keep[i]) {
    const float4 t = tgt_pts[match_pos[i]];
    v.x = t.x; v.y = t.y; v.z = t.z;
  }
  q[i] = v;
}
// impl/correspondence_estimation.hpp:265-266: drop if d_reciprocal > max^2 or the reciprocal NN is not the query
__global__ void recip_keep_kernel(const float4* __restrict__ cur, const int32_t* __restrict__ r_idx,
                                  const float* __restrict__ r_d2, uint32_t n, float max_d2, int use_max,
                                  uint8_t* __restrict__ keep) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && keep[i]) {
    const bool ok = r_idx[i] >= 0 && uint32_t(r_idx[i]) == __float_as_uint(cur[i].w) && !(use_max && r_d2[i] > max_d2);
    if (!ok) keep[i] = 0;
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

pclhip_status apply_correspondence_filters(pclhip_icp* icp, float max_d2, bool use_max) {
  pclhip_ctx* ctx = icp->ctx;
  hipStream_t s = ctx->stream;
  const uint32_t n = icp->n;
  icp->fetch_order = 0;
  if (n == 0) return PCLHIP_OK;
  const dim3 grid((n + TB - 1) / TB), block(TB);
  if (!icp->keep) PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &icp->keep, n));
  hipLaunchKernelGGL(rej_init_kernel, grid, block, 0, s, icp->match_pos, n, icp->keep);
  Guard g;
  g.ctx = ctx;
  unsigned int* d_cnt = nullptr;
  PCLHIP_CHECK_HIP(ctx, g.alloc(&d_cnt, sizeof(unsigned int)));

  if (icp->reciprocal) {
    // source index over the CURRENT (transformed) source, ids = original source indices (.w of cur)
    pclhip_index* src_ix = nullptr;
    pclhip_status st = build_index_from_float4(ctx, icp->src_cur, n, &src_ix);
    if (st != PCLHIP_OK) return st;
    float4* q = nullptr;
    int32_t* r_idx = nullptr;
    float* r_d2 = nullptr;
    hipError_t e1 = g.alloc(&q, size_t(n) * sizeof(float4)), e2 = g.alloc(&r_idx, size_t(n) * sizeof(int32_t)),
               e3 = g.alloc(&r_d2, size_t(n) * sizeof(float));
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
      pclhip_index_destroy(src_ix);
      set_error(ctx, "hipMalloc failed in the reciprocal filter");
      return PCLHIP_ERR_HIP;
    }
    hipLaunchKernelGGL(recip_query_kernel, grid, block, 0, s, icp->target->pts, icp->match_pos, icp->keep, n, q);
    st = launch_knn(src_ix, q, n, 1, r_idx, r_d2);  // rows = slots (q.w), results = original source ids
    if (st == PCLHIP_OK)
      hipLaunchKernelGGL(recip_keep_kernel, grid, block, 0, s, icp->src_cur, r_idx, r_d2, n, max_d2, use_max ? 1 : 0,
                         icp->keep);
    hipError_t e = hipStreamSynchronize(s);
    pclhip_index_destroy(src_ix);
    if (st != PCLHIP_OK) return st;
    PCLHIP_CHECK_HIP(ctx, e);
  }

  for (const pclhip_rejector& r : icp->rejectors) {
    switch (r.kind) {
      case PCLHIP_REJ_DISTANCE: {
        const float md = float(r.param);
        hipLaunchKernelGGL(rej_distance_kernel, grid, block, 0, s, icp->match_d2, n, md * md, icp->keep);
        break;
      }
      case PCLHIP_REJ_MEDIAN_DISTANCE: {
        uint32_t* k0 = nullptr;
        RsState* rs = nullptr;
        uint32_t* rs_hist = nullptr;
        PCLHIP_CHECK_HIP(ctx, g.alloc(&k0, size_t(n) * 4));
        PCLHIP_CHECK_HIP(ctx, g.alloc(&rs, sizeof(RsState)));
        PCLHIP_CHECK_HIP(ctx, g.alloc(&rs_hist, RS_BINS * sizeof(uint32_t)));
        PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(d_cnt, 0, sizeof(unsigned int), s));
        hipLaunchKernelGGL(rej_dist_key_kernel, grid, block, 0, s, icp->match_d2, icp->keep, n, k0, d_cnt);
        unsigned int cnt = 0;
        PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(&cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost, s));
        PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
        if (cnt == 0) break;
        uint32_t mbits = 0;  // dropped slots carry the largest key, so rank cnt / 2 counts kept distances only
        {
          const pclhip_status st = radix_select<uint32_t>(ctx, k0, n, cnt / 2, rs, rs_hist, &mbits);
          if (st != PCLHIP_OK) return st;
        }
        float mf;
        std::memcpy(&mf, &mbits, sizeof mf);
        icp->last_median = double(mf);  // nth_element at size/2 (:55-56)
        hipLaunchKernelGGL(rej_median_kernel, grid, block, 0, s, icp->match_d2, n, icp->last_median * r.param, icp->keep);
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
        break;
      }
      case PCLHIP_REJ_TRIMMED: {
        uint64_t* k0 = nullptr;
        PCLHIP_CHECK_HIP(ctx, g.alloc(&k0, size_t(n) * 8));
        PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(d_cnt, 0, sizeof(unsigned int), s));
        hipLaunchKernelGGL(rej_pair_key_kernel, grid, block, 0, s, icp->src_cur, icp->match_d2, icp->keep, n, k0, d_cnt);
        unsigned int cnt = 0;
        PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(&cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost, s));
        PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
        // :47-50 float product, floor, max with nr_min_correspondences_
        volatile float prod = float(r.param) * float(cnt);
        unsigned int nv = (unsigned int)std::floor(prod);
        if (nv < r.min_correspondences) nv = r.min_correspondences;
        if (nv < cnt) {
          if (nv == 0) {
            PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(icp->keep, 0, n, s));
          } else {
            RsState* rs = nullptr;
            uint32_t* rs_hist = nullptr;
            PCLHIP_CHECK_HIP(ctx, g.alloc(&rs, sizeof(RsState)));
            PCLHIP_CHECK_HIP(ctx, g.alloc(&rs_hist, RS_BINS * sizeof(uint32_t)));
            uint64_t thr = 0;  // the nv-th smallest (distance, query) key
            const pclhip_status st = radix_select<uint64_t>(ctx, k0, n, nv - 1, rs, rs_hist, &thr);
            if (st != PCLHIP_OK) return st;
            hipLaunchKernelGGL(rej_trim_kernel, grid, block, 0, s, icp->src_cur, icp->match_d2, n, thr, icp->keep);
          }
          icp->fetch_order = 2;
        }
        break;
      }
      default:
        set_error(ctx, "unknown rejector kind");
        return PCLHIP_ERR_INVALID;
    }
  }
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));  // scratch is released when `g` goes out of scope
  return PCLHIP_OK;
}

void preload_rejector_kernels() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(rs_pick_kernel));
}
}  // namespace pclhip
