// search.hip -- kernels that traverse the index: batched k-NN, fused k-NN + normal estimation, and
// the fused ICP iteration (transform -> 1-NN -> normal-system accumulation).
//
// Compiled with -ffp-contract=off: every float expression that has a counterpart in the reference
// keeps the reference's operation order and rounding (see traverse.hpp and the citations below).
#include <hip/hip_runtime.h>

#include <type_traits>

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "traverse.hpp"
#include "standoff.hpp"
#include "normals_math.hpp"
#include "device_scan.hpp"
#include "icp_xform.hpp"
#include <mutex>
#include <unordered_map>
#include <vector>

namespace pclhip {

constexpr int BLOCK = 256;
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;
constexpr uint32_t COLD_RUN = 4;  // groups per run of the stand-off search (icp_cold_search_body)

// =================================================================================================
// batched exact k-NN
// =================================================================================================
// Queries are Morton-ordered float4 (w = original query index, NO_INDEX bits for non-finite
// queries).  Results are written at the ORIGINAL query position: out[(orig*k) + c].
// No search here bounds leaves by discs: the box-only LDS layout (WaveLdsBoxT), for k = 1 with three KB of staging (NN1Min
// stages no original indices), and four waves per SIMD up to k = 8.  10M self-queries, k = 1: 1.80 -> 1.53 ms, from a
// stand-off 4.74 -> 4.20 ms; k = 8: 3.48 -> 3.41 and 6.40 -> 6.29 ms.
#define KNN_MINW(K) ((K) <= 8 ? 4 : 1)
template <int K>
using KnnWaveLds = WaveLdsBoxT<(K == 1 ? 3072 : LEAF_BATCH * LEAF_FLOATS * 4)>;
template <int K>
__global__ __launch_bounds__(BLOCK, KNN_MINW(K)) void knn_reg_kernel(IndexView ix, const float4* __restrict__ q,
                                                        uint32_t nq, int k, int32_t* __restrict__ out_idx,
                                                        float* __restrict__ out_d2, unsigned long long* gstats) {
  __shared__ KnnWaveLds<K> wl_s[WAVES_PER_BLOCK];
  __shared__ Box topbox_s[TOPCACHE_BOXES];
  load_top_cache(ix, topbox_s);
  const int lane = threadIdx.x & (WAVE - 1);
  const uint32_t ngroups = (nq + WAVE - 1) / WAVE;
  const GroupSchedule sched(ngroups);
  TraverseStats ts;
  GroupFeed feed(sched, ix.sched_ctr);
  for (uint32_t gl = feed.first(sched); gl != GroupFeed::END; gl = feed.advance()) {
    const uint32_t g = sched.global(gl);
    if (g >= ngroups) break;
    feed.ahead(gl);
    const uint32_t i = g * WAVE + lane;
    float4 p = make_float4(0, 0, 0, 0);
    bool valid = i < nq;
    if (valid) p = q[i];
    const uint32_t oq = __float_as_uint(p.w);
    const bool real = valid;  // has an output row
    valid = valid && isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
    const float qx[1] = {p.x}, qy[1] = {p.y}, qz[1] = {p.z};
    const bool vv[1] = {valid};
    if constexpr (K == 1) {
      NN1Min fast;
      fast.init(__builtin_inff());
      traverse<NN1Min, true>(ix, qx, qy, qz, vv, fast, wl_s[threadIdx.x / WAVE], topbox_s, ts);
      fast.resolve(ix, qx, qy, qz);
      NN1 pol;
      pol.soa = ix.soa;
      pol.key = KEY_NONE;
      pol.pos = fast.bestpos[0];
      if (fast.bestpos[0] != NO_INDEX) pol.key = make_key(fast.best[0], __float_as_uint(ix.pts[fast.bestpos[0]].w));
      // exactness: cross-leaf distance ties, or nothing below +inf although the index is not empty
      const bool redo[1] = {valid && (fast.tie[0] || fast.bestpos[0] == NO_INDEX)};
      if (__builtin_amdgcn_ballot_w64(redo[0]) != 0) {
        NN1 ex = pol;
        traverse<NN1, true>(ix, qx, qy, qz, redo, ex, wl_s[threadIdx.x / WAVE], topbox_s, ts);
        if (redo[0]) pol = ex;
      }
      if (real) {
        const uint32_t id = key_index(pol.key);
        out_idx[size_t(oq) * k] = (id == NO_INDEX) ? -1 : int32_t(id);
        out_d2[size_t(oq) * k] = key_dist(pol.key);
        for (int c = 1; c < k; ++c) {  // k > n
          out_idx[size_t(oq) * k + c] = -1;
          out_d2[size_t(oq) * k + c] = __builtin_inff();
        }
      }
    } else {
      TopKReg<K> pol;
      pol.init(KEY_NONE);
      traverse<TopKReg<K>, true>(ix, qx, qy, qz, vv, pol, wl_s[threadIdx.x / WAVE], topbox_s, ts);
      if (real) {
#pragma unroll
        for (int c = 0; c < K; ++c) {
          if (c < k) {
            const uint32_t id = key_index(pol.keys[c]);
            out_idx[size_t(oq) * k + c] = (id == NO_INDEX) ? -1 : int32_t(id);
            out_d2[size_t(oq) * k + c] = key_dist(pol.keys[c]);
          }
        }
      }
    }
  }
  flush_stats(ts, gstats);
}

__global__ __launch_bounds__(BLOCK) void knn_heap_kernel(IndexView ix, const float4* __restrict__ q,
                                                         uint32_t nq, int k, int32_t* __restrict__ out_idx,
                                                         float* __restrict__ out_d2, uint64_t* heap, unsigned long long* gstats) {
  __shared__ KnnWaveLds<64> wl_s[WAVES_PER_BLOCK];
  __shared__ Box topbox_s[TOPCACHE_BOXES];
  load_top_cache(ix, topbox_s);
  const int lane = threadIdx.x & (WAVE - 1);
  const uint32_t ngroups = (nq + WAVE - 1) / WAVE;
  const GroupSchedule sched(ngroups);
  TraverseStats ts;
  GroupFeed feed(sched, ix.sched_ctr);
  for (uint32_t gl = feed.first(sched); gl != GroupFeed::END; gl = feed.advance()) {
    const uint32_t g = sched.global(gl);
    if (g >= ngroups) break;
    feed.ahead(gl);
    const uint32_t i = g * WAVE + lane;
    float4 p = make_float4(0, 0, 0, 0);
    bool valid = i < nq;
    if (valid) p = q[i];
    const uint32_t oq = __float_as_uint(p.w);
    const bool real = valid;
    valid = valid && isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
    TopKHeap pol;
    pol.heap = heap + (real ? i : 0);
    pol.stride = nq;
    pol.k = real ? k : 0;
    pol.init(KEY_NONE);
    if (!valid) pol.root = 0;  // lanes without a finite query never insert (key < 0 is impossible)
    const float qx[1] = {p.x}, qy[1] = {p.y}, qz[1] = {p.z};
    const bool vv[1] = {valid};
    traverse(ix, qx, qy, qz, vv, pol, wl_s[threadIdx.x / WAVE], topbox_s, ts);
    if (real) {
      pol.sort_ascending();
      for (int c = 0; c < k; ++c) {
        const uint64_t key = pol.heap[size_t(c) * nq];
        const uint32_t id = key_index(key);
        out_idx[size_t(oq) * k + c] = (id == NO_INDEX) ? -1 : int32_t(id);
        out_d2[size_t(oq) * k + c] = key_dist(key);
      }
    }
  }
  flush_stats(ts, gstats);
}

static int persistent_blocks(pclhip_ctx* ctx, uint32_t ngroups, int blocks_per_cu) {
  const int64_t want = (int64_t(ngroups) + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  int64_t cap = int64_t(ctx->num_cus) * blocks_per_cu;
  if (cap < 8) cap = 8;
  int64_t b = want < cap ? want : cap;
  if (b < 1) b = 1;
  return int(b);
}

// Persistent grid = what is actually co-resident (occupancy query), so no block waits for a free
// slot while the others are already through their share of the groups.
template <class K>
static int resident_blocks(pclhip_ctx* ctx, K kernel, uint32_t ngroups) {
  // keyed by the kernel's address: instantiations with the same signature share this function
  static std::mutex mu;
  static std::unordered_map<const void*, int> cache;
  int per_cu;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(reinterpret_cast<const void*>(kernel));
    if (it == cache.end()) {
      int v = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, kernel, BLOCK, 0) != hipSuccess || v < 1) {
        (void)hipGetLastError();
        v = 2;
      }
      it = cache.emplace(reinterpret_cast<const void*>(kernel), v).first;
    }
    per_cu = it->second;
  }
  return persistent_blocks(ctx, ngroups, per_cu);
}

pclhip_status launch_knn(pclhip_index* ix, const float4* q_sorted, uint32_t nq, int k, int32_t* out_idx,
                         float* out_d2, bool timed) {
  pclhip_ctx* ctx = ix->ctx;
  if (nq == 0) return PCLHIP_OK;
  const uint32_t ngroups = (nq + WAVE - 1) / WAVE;
  const IndexView v = ix->view();
  hipStream_t s = ctx->stream;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timed) {
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
  }
  struct Timer {
    pclhip_index* ix; hipEvent_t a, b; hipStream_t s;
    ~Timer() {
      if (a == nullptr) return;  // an untimed launch: nothing to wait for
      (void)hipEventRecord(b, s);
      if (hipEventSynchronize(b) == hipSuccess) { float ms = 0; if (hipEventElapsedTime(&ms, a, b) == hipSuccess) ix->last_kernel_ms = ms; }
      (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    }
  } timer{ix, e0, e1, s};
  if (k == 1) {
    const int grid = resident_blocks(ctx, knn_reg_kernel<1>, ngroups);
    PCLHIP_LAUNCH_FED(ctx, knn_reg_kernel<1>, dim3(grid), dim3(BLOCK), 0, s, v, q_sorted, nq, k, out_idx, out_d2, ctx->stats);
  } else if (k <= 8) {
    const int grid = resident_blocks(ctx, knn_reg_kernel<8>, ngroups);
    PCLHIP_LAUNCH_FED(ctx, knn_reg_kernel<8>, dim3(grid), dim3(BLOCK), 0, s, v, q_sorted, nq, k, out_idx, out_d2, ctx->stats);
  } else if (k <= 16) {
    const int grid = resident_blocks(ctx, knn_reg_kernel<16>, ngroups);
    PCLHIP_LAUNCH_FED(ctx, knn_reg_kernel<16>, dim3(grid), dim3(BLOCK), 0, s, v, q_sorted, nq, k, out_idx, out_d2, ctx->stats);
  } else if (k <= 32) {
    const int grid = resident_blocks(ctx, knn_reg_kernel<32>, ngroups);
    PCLHIP_LAUNCH_FED(ctx, knn_reg_kernel<32>, dim3(grid), dim3(BLOCK), 0, s, v, q_sorted, nq, k, out_idx, out_d2, ctx->stats);
  } else {
    const size_t bytes = size_t(nq) * size_t(k) * sizeof(uint64_t);
    uint64_t* heap = nullptr;
    PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &heap, bytes));
    const int grid = resident_blocks(ctx, knn_heap_kernel, ngroups);
    PCLHIP_LAUNCH_FED(ctx, knn_heap_kernel, dim3(grid), dim3(BLOCK), 0, s, v, q_sorted, nq, k, out_idx, out_d2,
                       heap, ctx->stats);
    hipError_t e = hipStreamSynchronize(s);
    (void)dev_free(ctx, heap);
    PCLHIP_CHECK_HIP(ctx, e);
  }
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  return PCLHIP_OK;
}

// =================================================================================================
// fused k-NN + NormalEstimation over the indexed cloud itself
// =================================================================================================

template <int K>
__global__ __launch_bounds__(BLOCK, (K == 8 && PCLHIP_NRM_WAVES >= 4) ? 4 : 1) void normals_kernel(IndexView ix, int k, float vx, float vy, float vz,
                                                        float4* __restrict__ nrm_sorted,
                                                        unsigned long long* __restrict__ nan_count,
                                                        unsigned long long* gstats) {
  // k <= 8: 3 KB of staging (neither pass stages the original indices) + 3.75 KB of records per wave (TopKDist), which
  // the exact policy of tie lanes may overwrite with its fourth KB of staging -- the records are used up by then.  52 KB
  // per block: three blocks per CU (at 53 KB only two were resident: 3.9 against 2.6 ms)
  constexpr int REC_AT = 3072 / 4;  // first float of the records inside the staging buffer
  typedef typename std::conditional<PCLHIP_NRM_WAVES >= 4, WaveLdsBoxT<3072 + REC_BYTES>, WaveLdsT<3072 + REC_BYTES>>::type Nrm8Lds;
  typedef typename std::conditional<K == 8, Nrm8Lds, WaveLds>::type NrmWaveLds;
  __shared__ NrmWaveLds wl_s[WAVES_PER_BLOCK];
  __shared__ Box topbox_s[TOPCACHE_BOXES];
  load_top_cache(ix, topbox_s);
  const int lane = threadIdx.x & (WAVE - 1);
  const uint32_t ngroups = (ix.n + WAVE - 1) / WAVE;
  const GroupSchedule sched(ngroups);
  TraverseStats ts;
  const float qnan = __builtin_nanf("");
  GroupFeed feed(sched, ix.sched_ctr);
  for (uint32_t gl = feed.first(sched); gl != GroupFeed::END; gl = feed.advance()) {
    const uint32_t g = sched.global(gl);
    if (g >= ngroups) break;
    feed.ahead(gl);
    const uint32_t i = g * WAVE + lane;
    const bool valid = i < ix.n;
    float4 p = make_float4(0, 0, 0, 0);
    if (valid) p = ix.pts[i];
    TopKReg<K> pol;
    pol.init(KEY_NONE);
    const float qx[1] = {p.x}, qy[1] = {p.y}, qz[1] = {p.z};
    const bool vv[1] = {valid};
#ifdef PCLHIP_NRM_PROFILE  // -DPCLHIP_NRM_PROFILE: counters 5/6/7 become clock64() ticks of pass 1 / pass 2 / the plane fit
    uint64_t nrm_t = clock64();
#define NRM_LAP(i)                              \
  do {                                          \
    const uint64_t nrm_now = clock64();         \
    ts.c[i] += uint32_t(nrm_now - nrm_t);       \
    nrm_t = nrm_now;                            \
  } while (0)
#else
#define NRM_LAP(i) (void)0
#endif
    if constexpr (K == 8) {
      // Two passes instead of one top-K pass with identities.  (1) the k smallest DISTANCES (TopKDist: one
      // v_med3_f32 per slot and candidate), remembering per lane the leaves that contributed; (2) every lane reads
      // ITS remembered leaves again, straight from the index, and collects the positions of the candidates up to the
      // k-th distance (CollectLE) -- no second walk (it cost as much as the first: 33k against 36k ticks per group).
      // The <= k positions are then put into the reference's (distance, index) order by a 19-comparator network.
      // Exact distance ties at the k-th distance (more than k candidates collected) fall back to the one-pass exact
      // policy for that lane; a lane with more than REC_CAP contributing leaves sends its wave through the tight
      // second traversal this replaced.
      const int wave_id = threadIdx.x / WAVE;
      TopKDist<K> dist;
      uint32_t* const rec = reinterpret_cast<uint32_t*>(wl_s[wave_id].buf + REC_AT) + lane;
      dist.init(rec);
      const uint32_t own_leaf = uniform_u32((g * WAVE) / LEAF);
      static_assert(WAVE % LEAF == 0, "a wave's queries are whole leaves");
      if (valid) dist.seed_own_leaf(ix.soa, i / LEAF, qx, qy, qz);   // lanes i / LEAF = own_leaf ... own_leaf + 3
      traverse<TopKDist<K>, true>(ix, qx, qy, qz, vv, dist, wl_s[wave_id], topbox_s, ts, own_leaf);
      NRM_LAP(5);
      float thr = (k >= 1 && k <= K) ? dist.d[0] : 0.0f;
#pragma unroll
      for (int c = 1; c < K; ++c) thr = (c < k) ? dist.d[c] : thr;  // d[k - 1]
      [[maybe_unused]] const uint32_t nrec = valid ? dist.nrec : 0u;
#ifdef PCLHIP_NRM_STATS  // counters 5/6/7: waves with an overflowing lane, largest list of the wave, records of all lanes
      {
        uint32_t mx = nrec, sm = nrec;
        for (int o = 32; o > 0; o >>= 1) {
          mx = max(mx, uint32_t(__shfl_xor(int(mx), o)));
          sm += uint32_t(__shfl_xor(int(sm), o));
        }
        ts.c[5] += mx > REC_CAP ? 1u : 0u;
        ts.c[6] += mx;
        ts.c[7] += sm;
      }
#endif
      uint32_t cand[K];  // positions of the candidates up to the k-th distance, any order
      uint32_t ncand = 0;
#pragma unroll
      for (int c = 0; c < K; ++c) cand[c] = NO_INDEX;
#ifndef PCLHIP_NRM_REWALK  // A/B: -DPCLHIP_NRM_REWALK keeps the second traversal
      // (a lane without a finite k-th distance -- fewer than k points in reach -- has no threshold to compare against)
      if (__builtin_amdgcn_ballot_w64(valid && (nrec > REC_CAP || !(thr < __builtin_inff()))) == 0) {
        __builtin_amdgcn_wave_barrier();
        CollectCodes cc;
        cc.thr = thr;
        cc.cnt = 0;
        cc.codes = 0;
        // the records that still matter: nearest point not beyond the FINAL k-th distance
        uint32_t rel = 0;
        if constexpr (REC_MINS) {
#pragma unroll
          for (uint32_t r = 0; r < REC_CAP; ++r) {
            const float m = __uint_as_float(rec[(REC_MIN_ROW + r) * WAVE]);
            rel |= (r < nrec && m <= thr) ? (1u << r) : 0u;
          }
        } else {
          rel = (1u << nrec) - 1u;  // nrec <= REC_CAP here
        }
        uint32_t id = valid ? i / LEAF : NO_INDEX, row = REC_CAP;  // the own leaf (seed_own_leaf) first
        for (;;) {
          cc.leaf(ix.soa, id, row, qx, qy, qz);
          if (__builtin_amdgcn_ballot_w64(rel != 0) == 0) break;
          id = NO_INDEX;
          if (rel != 0) {
            row = uint32_t(__builtin_ctz(rel));
            id = rec[row * WAVE];
            rel &= rel - 1u;
          }
        }
        ncand = cc.cnt;
#pragma unroll
        for (int c = 0; c < K; ++c) {
          if (uint32_t(c) < ncand) {
            const uint32_t code = uint32_t(cc.codes >> (8 * c)) & 0xFFu;
            const uint32_t r = code >> 4;
            const uint32_t leaf = (r == REC_CAP) ? i / LEAF : rec[r * WAVE];
            cand[c] = leaf * LEAF + (code & 15u);
          }
        }
      } else
#endif
      {
        CollectLE<K> col;
        col.thr = thr;
        col.cnt = 0;
        col.over = false;
        col.list = rec;  // rows [0, 8) of the records: incomplete, or not needed
        traverse<CollectLE<K>, true>(ix, qx, qy, qz, vv, col, wl_s[wave_id], topbox_s, ts, own_leaf);
        __builtin_amdgcn_wave_barrier();
        ncand = col.cnt;
#pragma unroll
        for (int c = 0; c < K; ++c)
          if (uint32_t(c) < ncand) cand[c] = col.list[c * WAVE];
      }
      NRM_LAP(6);
      const bool redo[1] = {valid && ncand > uint32_t(K)};
      if (valid && ncand <= uint32_t(K)) {
        // (the candidates' points: all K gathers in flight together, empty slots re-read the query's own point -- see the
        //  covariance below)
        float4 tt[K];
#pragma unroll
        for (int c = 0; c < K; ++c) tt[c] = ix.pts[uint32_t(c) < ncand ? cand[c] : i];
#pragma unroll
        for (int c = 0; c < K; ++c) {
          if (uint32_t(c) < ncand) {
            const float4 t = tt[c];
            pol.keys[c] = make_key(l2_simple(p.x, p.y, p.z, t.x, t.y, t.z), __float_as_uint(t.w));
            pol.pos[c] = cand[c];
          }
        }
        // Batcher's odd-even merge sort for 8 keys (19 comparators): ascending (distance, index), empty slots last
        const auto cx = [&](int a, int b) {
          const bool sw = pol.keys[b] < pol.keys[a];
          const uint64_t ka = pol.keys[a], kb = pol.keys[b];
          const uint32_t pa = pol.pos[a], pb = pol.pos[b];
          pol.keys[a] = sw ? kb : ka; pol.keys[b] = sw ? ka : kb;
          pol.pos[a] = sw ? pb : pa; pol.pos[b] = sw ? pa : pb;
        };
        cx(0, 1); cx(2, 3); cx(4, 5); cx(6, 7);
        cx(0, 2); cx(1, 3); cx(4, 6); cx(5, 7);
        cx(1, 2); cx(5, 6);
        cx(0, 4); cx(1, 5); cx(2, 6); cx(3, 7);
        cx(2, 4); cx(3, 5);
        cx(1, 2); cx(3, 4); cx(5, 6);
      }
      if (__builtin_amdgcn_ballot_w64(redo[0]) != 0) {  // ties at the k-th distance: the exact one-pass policy
        TopKReg<K> ex;
        ex.init(KEY_NONE);
        traverse<TopKReg<K>, true>(ix, qx, qy, qz, redo, ex, wl_s[wave_id], topbox_s, ts);
        if (redo[0]) pol = ex;
      }
    } else {
      traverse<TopKReg<K>, true>(ix, qx, qy, qz, vv, pol, wl_s[threadIdx.x / WAVE], topbox_s, ts);
    }
    if (valid) {
      // normal_3d.hpp:59-66 + normal_3d.h:308-322: fewer than 3 neighbours -> NaN
      int found = 0;
#pragma unroll
      for (int c = 0; c < K; ++c)
        if (c < k && pol.pos[c] != NO_INDEX) ++found;
      float4 out;
      if (found < 3) {
        out = make_float4(qnan, qnan, qnan, qnan);
        atomicAdd(nan_count, 1ull);
      } else {
        Cov cv;
        // the neighbours' coordinates: up to GB gathers in flight together (a gather under `c < found` is waited for where it
        // stands -- K memory round trips one after the other per wavefront); slots past `found` re-read neighbour 0 and are
        // not added.  The sums take the neighbours in the order 0, 1, ... as before.
        constexpr int GB = K < 8 ? K : 8;
        float4 pc[GB];
#pragma unroll
        for (int c0 = 0; c0 < K; c0 += GB) {
#pragma unroll
          for (int f = 0; f < GB; ++f) {
            const int c = c0 + f;
            pc[f] = ix.pts[(c < K && c < found) ? pol.pos[c < K ? c : 0] : pol.pos[0]];
          }
#pragma unroll
          for (int f = 0; f < GB; ++f) {
            const int c = c0 + f;
            if (c == 0) cv.start(pc[0].x, pc[0].y, pc[0].z);
            if (c < K && c < found) cv.add(pc[f].x, pc[f].y, pc[f].z);
          }
        }
        float cov[9];
        cv.finish(found, cov);
        float nx, ny, nz, curv;
        solve_plane(cov, nx, ny, nz, curv);
        flip_to_viewpoint(p.x, p.y, p.z, vx, vy, vz, nx, ny, nz);
        out = make_float4(nx, ny, nz, curv);
      }
      nrm_sorted[i] = out;
    }
    NRM_LAP(7);
  }
#undef NRM_LAP
  flush_stats(ts, gstats);
}

// k > 32: neighbours through the generic k-NN kernel (results by original index) then this pass.
__global__ __launch_bounds__(BLOCK) void normals_from_knn_kernel(IndexView ix, int k, const uint32_t* __restrict__ rank,
                                                                 const int32_t* __restrict__ knn_by_sorted, float vx,
                                                                 float vy, float vz, float4* __restrict__ nrm_sorted,
                                                                 unsigned long long* __restrict__ nan_count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ix.n) return;
  const float4 p = ix.pts[i];
  const int32_t* nb = knn_by_sorted + size_t(i) * k;
  int found = 0;
  for (int c = 0; c < k; ++c) found += (nb[c] >= 0);
  const float qnan = __builtin_nanf("");
  float4 out;
  if (found < 3) {
    out = make_float4(qnan, qnan, qnan, qnan);
    atomicAdd(nan_count, 1ull);
  } else {
    Cov cv;
    const float4 p0 = ix.pts[rank[nb[0]]];
    cv.start(p0.x, p0.y, p0.z);
    for (int c = 0; c < found; ++c) {
      const float4 pc = ix.pts[rank[nb[c]]];
      cv.add(pc.x, pc.y, pc.z);
    }
    float cov[9];
    cv.finish(found, cov);
    float nx, ny, nz, curv;
    solve_plane(cov, nx, ny, nz, curv);
    flip_to_viewpoint(p.x, p.y, p.z, vx, vy, vz, nx, ny, nz);
    out = make_float4(nx, ny, nz, curv);
  }
  nrm_sorted[i] = out;
}

// Feature::setSearchSurface / setIndices (features/include/pcl/features/impl/feature.hpp:104-118): the index is the
// SURFACE, the queries are another cloud.  Row j of `knn` holds the original surface ids of query j's neighbours in
// the order the search returns them; the plane is fitted to those surface points (normal_3d.h:308-322) and the
// normal flipped towards the viewpoint as seen from the QUERY point (impl/normal_3d.hpp:66,87).
__global__ __launch_bounds__(BLOCK) void normals_at_kernel(IndexView ix, int k, const uint32_t* __restrict__ rank,
                                                           const float4* __restrict__ q, uint32_t nq,
                                                           const int32_t* __restrict__ knn, float vx, float vy, float vz,
                                                           float4* __restrict__ out,
                                                           unsigned long long* __restrict__ nan_count) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nq) return;
  const float4 p = q[j];
  const int32_t* nb = knn + size_t(j) * k;
  int found = 0;
  for (int c = 0; c < k; ++c) found += (nb[c] >= 0);
  const float qnan = __builtin_nanf("");
  float4 o;
  if (found < 3 || !(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) {
    o = make_float4(qnan, qnan, qnan, qnan);
    atomicAdd(nan_count, 1ull);
  } else {
    Cov cv;
    const float4 p0 = ix.pts[rank[nb[0]]];
    cv.start(p0.x, p0.y, p0.z);
    for (int c = 0; c < found; ++c) {
      const float4 pc = ix.pts[rank[nb[c]]];
      cv.add(pc.x, pc.y, pc.z);
    }
    float cov[9];
    cv.finish(found, cov);
    float nx, ny, nz, curv;
    solve_plane(cov, nx, ny, nz, curv);
    flip_to_viewpoint(p.x, p.y, p.z, vx, vy, vz, nx, ny, nz);
    o = make_float4(nx, ny, nz, curv);
  }
  out[j] = o;
}

__global__ void iota_w_kernel(const float4* __restrict__ pts, uint32_t n, float4* __restrict__ q) {
  // queries = the sorted points with w = SORTED position, so knn results land at row `sorted pos`
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float4 p = pts[i];
    p.w = __uint_as_float(i);
    q[i] = p;
  }
}

pclhip_status launch_normals(pclhip_index* ix, int k, const float vp[3], uint64_t* nan_count) {
  pclhip_ctx* ctx = ix->ctx;
  hipStream_t s = ctx->stream;
  DeviceScope scope(ctx);
  if (!ix->nrm) PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &ix->nrm, size_t(ix->n_pad > 0 ? ix->n_pad : 1) * sizeof(float4)));
  unsigned long long* d_nan = nullptr;
  PCLHIP_CHECK_HIP(ctx, scope.alloc(&d_nan, sizeof(unsigned long long)));
  PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(d_nan, 0, sizeof(unsigned long long), s));
  PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(ix->nrm, 0xFF, size_t(ix->n_pad) * sizeof(float4), s));  // NaN pads
  const IndexView v = ix->view();
  const uint32_t ngroups = (ix->n + WAVE - 1) / WAVE;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  PCLHIP_CHECK_HIP(ctx, scope.event(&e0));
  PCLHIP_CHECK_HIP(ctx, scope.event(&e1));
  (void)hipEventRecord(e0, s);
  if (ix->n > 0) {
    if (k <= 8) {
      PCLHIP_LAUNCH_FED(ctx, normals_kernel<8>, dim3(resident_blocks(ctx, normals_kernel<8>, ngroups)), dim3(BLOCK), 0, s, v, k, vp[0],
                         vp[1], vp[2], ix->nrm, d_nan, ctx->stats);
    } else if (k <= 16) {
      PCLHIP_LAUNCH_FED(ctx, normals_kernel<16>, dim3(resident_blocks(ctx, normals_kernel<16>, ngroups)), dim3(BLOCK), 0, s, v, k, vp[0],
                         vp[1], vp[2], ix->nrm, d_nan, ctx->stats);
    } else if (k <= 32) {
      PCLHIP_LAUNCH_FED(ctx, normals_kernel<32>, dim3(resident_blocks(ctx, normals_kernel<32>, ngroups)), dim3(BLOCK), 0, s, v, k, vp[0],
                         vp[1], vp[2], ix->nrm, d_nan, ctx->stats);
    } else {  // k > 32: materialise the k-NN lists (heap kernel), then fit the planes
      float4* q = nullptr;
      int32_t* nb = nullptr;
      float* nd = nullptr;
      PCLHIP_CHECK_HIP(ctx, scope.alloc(&q, size_t(ix->n) * sizeof(float4)));
      PCLHIP_CHECK_HIP(ctx, scope.alloc(&nb, size_t(ix->n) * k * sizeof(int32_t)));
      PCLHIP_CHECK_HIP(ctx, scope.alloc(&nd, size_t(ix->n) * k * sizeof(float)));
      hipLaunchKernelGGL(iota_w_kernel, dim3((ix->n + 255) / 256), dim3(256), 0, s, ix->pts, ix->n, q);
      const pclhip_status st = launch_knn(ix, q, ix->n, k, nb, nd);
      if (st != PCLHIP_OK) {
        (void)hipStreamSynchronize(s);  // nothing may still use the scope's buffers when they are freed
        return st;
      }
      hipLaunchKernelGGL(normals_from_knn_kernel, dim3((ix->n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, v, k, ix->rank, nb,
                         vp[0], vp[1], vp[2], ix->nrm, d_nan);
    }
  }
  (void)hipEventRecord(e1, s);
  unsigned long long h = 0;
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(&h, d_nan, sizeof h, hipMemcpyDeviceToHost, s);
  const hipError_t es = hipStreamSynchronize(s);  // always drain the stream before the scope frees its buffers
  PCLHIP_CHECK_HIP(ctx, e);
  PCLHIP_CHECK_HIP(ctx, es);
  float ms = 0;
  if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) ix->last_kernel_ms = ms;
  if (nan_count) *nan_count = h;
  ix->has_normals = true;
  return PCLHIP_OK;
}

// queries: `nq` dense float4 records in slot order (device); out: one float4 (normal, curvature) per slot
pclhip_status launch_normals_at(pclhip_index* ix, const float4* queries, uint32_t nq, int k, double radius, const float vp[3],
                                float4* out, uint64_t* nan_count) {
  pclhip_ctx* ctx = ix->ctx;
  hipStream_t s = ctx->stream;
  if (nan_count) *nan_count = 0;
  if (nq == 0) return PCLHIP_OK;
  DeviceScope scope(ctx);
  float4* qs = nullptr;  // the queries in kd order (compact 64-query groups), w = slot
  PCLHIP_CHECK_HIP(ctx, scope.alloc(&qs, size_t(nq) * sizeof(float4)));
  uint32_t nf = 0;
  float lo[3], hi[3];
  pclhip_status st = spatial_order(ctx, queries, sizeof(float4), nq, nullptr, 0, qs, nq, &nf, lo, hi, true, nullptr);
  if (st != PCLHIP_OK) return st;
  if (k < 1) return launch_normals_radius_at(ix, qs, nq, radius, vp, out, nan_count);
  unsigned long long* d_nan = nullptr;
  int32_t* nb = nullptr;
  float* nd = nullptr;
  PCLHIP_CHECK_HIP(ctx, scope.alloc(&d_nan, sizeof(unsigned long long)));
  PCLHIP_CHECK_HIP(ctx, scope.alloc(&nb, (size_t(nq) + 1) * k * sizeof(int32_t)));   // + the dump row of a sparse layout's padding
  PCLHIP_CHECK_HIP(ctx, scope.alloc(&nd, (size_t(nq) + 1) * k * sizeof(float)));
  PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(d_nan, 0, sizeof(unsigned long long), s));
  // few queries against a large surface: fewer of them per wavefront (api.hip: sparse_layout)
  const float4* q_run = qs;
  uint32_t nq_run = nq;
  {
    float4* qe = nullptr;
    st = sparse_layout(ctx, qs, nq, ix->n, &qe, &nq_run);
    if (st != PCLHIP_OK) return st;
    if (qe != nullptr) {
      scope.mem.push_back(qe);
      q_run = qe;
    }
  }
  st = launch_knn(ix, q_run, nq_run, k, nb, nd);
  if (st != PCLHIP_OK) {
    (void)hipStreamSynchronize(s);  // nothing may still use the scope's buffers when they are freed
    return st;
  }
  hipLaunchKernelGGL(normals_at_kernel, dim3((nq + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, ix->view(), k, ix->rank, queries, nq,
                     nb, vp[0], vp[1], vp[2], out, d_nan);
  unsigned long long h = 0;
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(&h, d_nan, sizeof h, hipMemcpyDeviceToHost, s);
  const hipError_t es = hipStreamSynchronize(s);  // always drain the stream before the scope frees its buffers
  PCLHIP_CHECK_HIP(ctx, e);
  PCLHIP_CHECK_HIP(ctx, es);
  if (nan_count) *nan_count = h;
  return PCLHIP_OK;
}

// =================================================================================================
// GeneralizedIterativeClosestPoint::computeCovariances (registration/include/pcl/registration/impl/gicp.hpp
// :70-147): per point the k nearest neighbours (k <= 32), their covariance in double about the query
// point (float differences, double sums), and the covariance "regularised" to singular values
// (1, 1, epsilon): U diag(1,1,eps) U^T = I - (1 - eps) n n^T with n the singular vector of the smallest
// singular value.  One 3x3 row-major double matrix per sorted point.
// =================================================================================================
__device__ __forceinline__ void smallest_eigenvector3(double A[3][3], double n[3]) {
  // cyclic Jacobi on the symmetric 3x3; V accumulates the rotations
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    if (off < 1e-300) break;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - sn * akq;
          A[k][q] = sn * akp + c * akq;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - sn * aqk;
          A[q][k] = sn * apk + c * aqk;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - sn * vkq;
          V[k][q] = sn * vkp + c * vkq;
        }
      }
    }
  }
  // singular values of a symmetric matrix are |eigenvalues|
  const double w0 = fabs(A[0][0]), w1 = fabs(A[1][1]), w2 = fabs(A[2][2]);
  const int m = (w0 <= w1 && w0 <= w2) ? 0 : ((w1 <= w2) ? 1 : 2);
  n[0] = V[0][m];
  n[1] = V[1][m];
  n[2] = V[2][m];
}

__global__ __launch_bounds__(BLOCK) void gicp_cov_kernel(IndexView ix, int k, double eps, double* __restrict__ cov_sorted,
                                                         unsigned long long* gstats) {
  __shared__ WaveLds wl_s[WAVES_PER_BLOCK];
  __shared__ Box topbox_s[TOPCACHE_BOXES];
  load_top_cache(ix, topbox_s);
  const int lane = threadIdx.x & (WAVE - 1);
  const uint32_t ngroups = (ix.n + WAVE - 1) / WAVE;
  const GroupSchedule sched(ngroups);
  TraverseStats ts;
  GroupFeed feed(sched, ix.sched_ctr);
  for (uint32_t gl = feed.first(sched); gl != GroupFeed::END; gl = feed.advance()) {
    const uint32_t g = sched.global(gl);
    if (g >= ngroups) break;
    feed.ahead(gl);
    const uint32_t i = g * WAVE + lane;
    const bool valid = i < ix.n;
    float4 p = make_float4(0, 0, 0, 0);
    if (valid) p = ix.pts[i];
    TopKReg<32> pol;
    pol.init(KEY_NONE);
    {
      const float qx[1] = {p.x}, qy[1] = {p.y}, qz[1] = {p.z};
      const bool vv[1] = {valid};
      traverse<TopKReg<32>, true>(ix, qx, qy, qz, vv, pol, wl_s[threadIdx.x / WAVE], topbox_s, ts);
    }
    if (valid) {
      double mean[3] = {0, 0, 0}, c00 = 0, c10 = 0, c11 = 0, c20 = 0, c21 = 0, c22 = 0;
      // eight neighbours per batch, their gathers in flight together (slots from k on re-read neighbour 0 and are skipped);
      // the sums take the neighbours in the order 0, 1, ... as before
#pragma unroll
      for (int c0 = 0; c0 < 32; c0 += 8) {
        if (c0 >= k) break;
        float4 qq[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) qq[f] = ix.pts[pol.pos[(c0 + f) < k ? c0 + f : 0]];
#pragma unroll
        for (int f = 0; f < 8; ++f) {
          if (c0 + f < k) {  // the index holds >= k points (checked by the caller): all k slots are filled
            const float4 q = qq[f];
            const double ptx = double(__fsub_rn(q.x, p.x)), pty = double(__fsub_rn(q.y, p.y)), ptz = double(__fsub_rn(q.z, p.z));
            mean[0] += ptx; mean[1] += pty; mean[2] += ptz;
            c00 += ptx * ptx;
            c10 += pty * ptx; c11 += pty * pty;
            c20 += ptz * ptx; c21 += ptz * pty; c22 += ptz * ptz;
          }
        }
      }
      const double kk = double(k);
      mean[0] /= kk; mean[1] /= kk; mean[2] /= kk;
      double A[3][3];
      A[0][0] = c00 / kk - mean[0] * mean[0];
      A[1][0] = A[0][1] = c10 / kk - mean[1] * mean[0];
      A[1][1] = c11 / kk - mean[1] * mean[1];
      A[2][0] = A[0][2] = c20 / kk - mean[2] * mean[0];
      A[2][1] = A[1][2] = c21 / kk - mean[2] * mean[1];
      A[2][2] = c22 / kk - mean[2] * mean[2];
      double n[3];
      smallest_eigenvector3(A, n);
      double* o = cov_sorted + size_t(i) * 9;
      const double f = 1.0 - eps;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) o[3 * r + c] = (r == c ? 1.0 : 0.0) - f * n[r] * n[c];
    }
  }
  flush_stats(ts, gstats);
}

pclhip_status launch_gicp_covariances(pclhip_index* ix, int k, double eps, double* cov_sorted) {
  pclhip_ctx* ctx = ix->ctx;
  if (ix->n == 0) return PCLHIP_OK;
  const IndexView v = ix->view();
  const uint32_t ngroups = (ix->n + WAVE - 1) / WAVE;
  PCLHIP_LAUNCH_FED(ctx, gicp_cov_kernel, dim3(resident_blocks(ctx, gicp_cov_kernel, ngroups)), dim3(BLOCK), 0, ctx->stream, v, k,
                     eps, cov_sorted, ctx->stats);
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  return PCLHIP_OK;
}

// =================================================================================================
// ICP iteration
// =================================================================================================
// Mat34, xform_row, in_region: icp_xform.hpp (shared with the per-lane search of lane.hip)

constexpr int NS = PCLHIP_ICP_NSUMS;
#ifndef PCLHIP_DEEP_FROM
#define PCLHIP_DEEP_FROM 12   // groups per wave from which the seeded body's pipeline runs two groups deep (icp_search_body)
#endif

// Queries per wavefront of the ICP search kernels: 64 >> s, s in bits 8-10 of `flags` (0: the full wavefront).
__host__ __device__ inline uint32_t search_fill_of(int flags) { return uint32_t(WAVE) >> ((uint32_t(flags) >> 8) & 7u); }

// -------------------------------------------------------------------------------------------------
// The iteration is two kernels: a search-only kernel without the 27 fp64 accumulators of the plane system (fewer
// registers, room to software-pipeline the next group's loads) followed by a streaming accumulate kernel.  (Round 1's
// single-kernel form was measured slower and left the library in round 4.)
// With `ctl` (device-driven loop, icp_loop.hip) the transform, the "alignment starts here" flag and the stop
// flag come from device memory, written by the icp_solve_kernel of the previous iteration: iterations are
// queued back to back and the host never sits between them.  A starting alignment reads the pristine
// source `src0` instead of the working copy and has no seeds, so no reset copy is needed either.

// no policy of the ICP search kernels stages the w chunks: 3 KB of staging per wave (see WaveLdsT)
typedef WaveLdsT<3072> IcpWaveLds;

// OWNED (target sharding in the device-driven loop, pclhip_internal.hpp: OwnedGroups): the groups come from the launch's list
// of served groups, and a group whose working copy missed some launches is brought up to date first.
struct NoOwnedGroups {};
// the working copy of point p after the launches [from, to) it missed: the very operations those launches would have applied
__device__ __forceinline__ void own_replay(const OwnedState* st, uint32_t from, uint32_t to, int order, float4& p) {
  for (uint32_t e = from; e < to; ++e) {
    const float* m = st->hist[e];
    const float x = xform_row(m[0], m[1], m[2], m[3], p.x, p.y, p.z, order);
    const float y = xform_row(m[4], m[5], m[6], m[7], p.x, p.y, p.z, order);
    const float z = xform_row(m[8], m[9], m[10], m[11], p.x, p.y, p.z, order);
    p.x = x; p.y = y; p.z = z;
  }
}

template <int Q, bool SPARSE, bool OWNED = false, class OG = NoOwnedGroups, bool DEEP = false>
__device__ __forceinline__ void icp_search_body(const IndexView& ix, float4* __restrict__ cur,
                                                const float4* __restrict__ src0, uint32_t ns, Mat34 T,
                                                const IcpControl* __restrict__ ctl, const RegionBox& region, int order,
                                                float bound, int flags, uint32_t* __restrict__ match_pos,
                                                uint32_t* __restrict__ match, float* __restrict__ match_d2,
                                                unsigned long long* gstats, IcpWaveLds* wl_s, Box* topbox_s,
                                                const OG& og = OG()) {
  static_assert(!OWNED || Q == 1, "served-group lists are per 64-point group");
  bool restart = false;
  if (ctl != nullptr) {
    if (ctl->stop != 0) return;  // the alignment ended before this (speculatively queued) launch
    restart = ctl->restart != 0;
    if ((flags & 4) != 0 && !restart) return;  // SEARCH_RESTART_ONLY: the seeded launches are lane.hip's
#pragma unroll
    for (int i = 0; i < 12; ++i) T.m[i] = ctl->T_apply[i];
  }
  const float4* in = restart ? src0 : cur;  // may alias cur (written below, other groups only)
  const int use_max = flags & 1;        // a finite max correspondence distance is set
  load_top_cache(ix, topbox_s);
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = threadIdx.x / WAVE;
  // queries per wavefront: Q per lane -- or FEWER (flags bits 8-10, search_fill_of): a source that is sparse against the target
  // index (a small scan against a large map) gets 64 >> s consecutive queries per wavefront, in its first lanes
  const uint32_t GROUP = (Q == 1 && !OWNED) ? search_fill_of(flags) : uint32_t(WAVE * Q);
  const bool lane_in_group = (Q != 1 || OWNED) ? true : uint32_t(threadIdx.x & (WAVE - 1)) < GROUP;
  uint32_t ngroups = (ns + GROUP - 1) / GROUP;
  uint32_t epoch = 0;
  if constexpr (OWNED) {
    ngroups = og.count[0];  // slots of the served list; gid() maps a slot to its group
    epoch = og.state->epoch;
  }
  const auto gid = [&](uint32_t slot) -> uint32_t {
    if constexpr (OWNED) return slot < ngroups ? og.list[slot] : 0u;
    else return slot;
  };
  const GroupSchedule sched(ngroups);
  TraverseStats ts;
  // software pipeline, two groups deep: while group g is searched, the points AND the seed target points of g + 1 are in
  // flight (its seed positions arrived during g - 1), and so are the seed positions of g + 2 -- a group finds everything
  // it starts with in registers.  (Until round 6 the seed points of g + 1 were asked for after g's traversal, when its seed
  // positions had just been looked at: a full memory round trip at the head of every group.)  The feed therefore runs two
  // groups ahead: the ticket asked at the top of g is read at the bottom of g as g + 3.
  // Only where a wave has groups enough (`deep`): a wave that runs two groups ahead also HOLDS two groups while the launch
  // drains, and with four groups per wave (2^20 points) that imbalance costs more than the round trip (config 2, same box:
  // 0.1565 ms per step one group ahead, 0.1675 two ahead) -- small launches keep the pipeline of rounds 1-5.
  uint32_t gl = sched.first();
  uint32_t g = (gl < sched.end()) ? sched.global(gl) : ngroups;
  GroupFeed feed(sched, ix.sched_ctr);
  const auto next_now = [&](uint32_t from) -> uint32_t {   // the group after `from`, waiting for the counter if need be
    uint32_t out = GroupFeed::END;
    if (from == GroupFeed::END) return out;
    if (!feed.static_next(from, out)) {
      feed.request();
      out = feed.resolve();
    }
    return out;
  };
  constexpr bool deep = DEEP;   // chosen by the host from the launch's groups per wave (icp_deep_pipeline): one body per depth
  uint32_t gl_next = (gl < sched.end()) ? next_now(gl) : GroupFeed::END;
  uint32_t gl_nn = deep ? next_now(gl_next) : GroupFeed::END;
  float4 p_n[Q], t_n[Q];
  uint32_t sp_n[Q], sp_nn[Q];
  uint32_t gid_n = gid(g), st_n = 0;  // the group behind slot g; how many launches its working copy has seen
  if constexpr (OWNED) st_n = (g < ngroups && !restart) ? (og.stamp[gid_n] & 0x7FFFFFFFu) : 0u;
  {
    const uint32_t g1 = (gl_next != GroupFeed::END) ? sched.global(gl_next) : ngroups;
    const uint32_t gid1 = gid(g1);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      p_n[q] = make_float4(0, 0, 0, 0);
      t_n[q] = make_float4(0, 0, 0, 0);
      sp_n[q] = NO_INDEX;
      sp_nn[q] = NO_INDEX;
      const uint32_t i = gid_n * GROUP + q * WAVE + lane;
      if (g < ngroups && i < ns && lane_in_group) {
        p_n[q] = (OWNED && st_n == 0u) ? src0[i] : in[i];
        sp_n[q] = restart ? NO_INDEX : match_pos[i];
      }
      const uint32_t i1 = gid1 * GROUP + q * WAVE + lane;
      if (g1 < ngroups && i1 < ns && lane_in_group) sp_nn[q] = restart ? NO_INDEX : match_pos[i1];
      if (sp_n[q] != NO_INDEX) t_n[q] = ix.pts[sp_n[q]];
    }
  }
#ifdef PCLHIP_ICP_PROFILE  // counters 5/6/7 become clock64() ticks: up to the traversal / the traversal / after it
  uint64_t icp_t = clock64();
#define ICP_LAP(i)                              \
  do {                                          \
    const uint64_t icp_now = clock64();         \
    ts.c[i] += uint32_t(icp_now - icp_t);       \
    icp_t = icp_now;                            \
  } while (0)
#else
#define ICP_LAP(i) (void)0
#endif
  while (g < ngroups) {
    float4 p[Q], t0[Q];
    uint32_t seed_pos[Q];
    bool in_range[Q], valid[Q];
    float qx[Q], qy[Q], qz[Q];
    const uint32_t gcur = gid_n, st_cur = st_n;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      p[q] = p_n[q];
      t0[q] = t_n[q];
      seed_pos[q] = sp_n[q];
      in_range[q] = lane_in_group && (gcur * GROUP + q * WAVE + lane) < ns;
    }
    // next group (g2): its points and -- its seed positions are here since the previous group -- its seed target points;
    // the group after it (g3): its seed positions
    const uint32_t g2 = (gl_next != GroupFeed::END) ? sched.global(gl_next) : ngroups;
    const uint32_t g3 = (gl_nn != GroupFeed::END) ? sched.global(gl_nn) : ngroups;
    gid_n = gid(g2);
    const uint32_t gid_nn = gid(g3);
    if constexpr (OWNED) st_n = (g2 < ngroups && !restart) ? (og.stamp[gid_n] & 0x7FFFFFFFu) : 0u;
    uint32_t gl_after = GroupFeed::END;
    bool asked = false;
    const uint32_t gl_last = deep ? gl_nn : gl_next;   // the last group this wave knows of
    if (gl_last != GroupFeed::END && !feed.static_next(gl_last, gl_after)) {
      feed.request();
      asked = true;
    }
    bool next_ok[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const uint32_t i2 = gid_n * GROUP + q * WAVE + lane;
      next_ok[q] = g2 < ngroups && i2 < ns && lane_in_group;
      p_n[q] = make_float4(0, 0, 0, 0);
      t_n[q] = make_float4(0, 0, 0, 0);
      if (deep) {
        sp_n[q] = sp_nn[q];   // (NO_INDEX where the lane has no point in g2: it was loaded under the same conditions)
        if (next_ok[q]) {
          p_n[q] = (OWNED && st_n == 0u) ? src0[i2] : in[i2];
          if (sp_n[q] != NO_INDEX) t_n[q] = ix.pts[sp_n[q]];
        }
        const uint32_t i3 = gid_nn * GROUP + q * WAVE + lane;
        sp_nn[q] = NO_INDEX;
        if (g3 < ngroups && i3 < ns && lane_in_group) sp_nn[q] = restart ? NO_INDEX : match_pos[i3];
      } else {
        sp_n[q] = NO_INDEX;
        if (next_ok[q]) {
          p_n[q] = (OWNED && st_n == 0u) ? src0[i2] : in[i2];
          sp_n[q] = restart ? NO_INDEX : match_pos[i2];
        }
      }
    }
    NN1MinT<Q> fast;
    fast.init(bound);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      valid[q] = in_range[q] && isfinite(p[q].x) && isfinite(p[q].y) && isfinite(p[q].z);
      if (valid[q]) {
        if constexpr (OWNED) own_replay(og.state, st_cur, restart ? 0u : epoch, order, p[q]);  // the launches it was not served in
        const float x = xform_row(T.m[0], T.m[1], T.m[2], T.m[3], p[q].x, p[q].y, p[q].z, order);
        const float y = xform_row(T.m[4], T.m[5], T.m[6], T.m[7], p[q].x, p[q].y, p[q].z, order);
        const float z = xform_row(T.m[8], T.m[9], T.m[10], T.m[11], p[q].x, p[q].y, p[q].z, order);
        p[q].x = x; p[q].y = y; p[q].z = z;
        cur[gcur * GROUP + q * WAVE + lane] = p[q];
        // target sharding: every rank moves the whole source, but serves only the points inside its region
        if (region.on) valid[q] = in_region(region, x, y, z);
        if (valid[q] && seed_pos[q] != NO_INDEX)
          fast.seed(q, l2_simple(x, y, z, t0[q].x, t0[q].y, t0[q].z), seed_pos[q]);
      } else if (in_range[q] && (restart || (OWNED && st_cur == 0u))) {
        cur[gcur * GROUP + q * WAVE + lane] = p[q];  // non-finite points travel unchanged (icp.hpp:97-98)
      }
      qx[q] = p[q].x; qy[q] = p[q].y; qz[q] = p[q].z;
    }

    // hint for the descent: the leaf of one lane's seed (any lane: the containment test inside traverse()
    // decides whether the shortcut is valid for the whole wave)
    uint32_t start_leaf = NO_INDEX;
    const uint64_t hm = __builtin_amdgcn_ballot_w64(valid[0] && seed_pos[0] != NO_INDEX);
    if ((flags & 2) && hm != 0) start_leaf = uint32_t(__builtin_amdgcn_readlane(int(seed_pos[0]), __builtin_ctzll(hm))) / LEAF;
    // Every seed of the group is far from its query -- the second iteration of an alignment, whose first transform slid
    // the queries tens of point spacings along the surface: one fresh seed for the group (a point next to its centre,
    // found by one greedy walk of the wave) gives every lane a radius of a few spacings instead of tens.
    if constexpr (Q == 1 && !OWNED) {
      if ((flags & 8) != 0 && hm != 0 && ix.disc != nullptr) {  // SEARCH_RESEED
        const float INF = __builtin_inff(), BIG = 3.402823466e+38f;
        const float near = wave_min_f((valid[0] && seed_pos[0] != NO_INDEX) ? fast.best[0] : INF);
        if (near > ix.disc_from) {
          float lx = valid[0] ? qx[0] : BIG, ly = valid[0] ? qy[0] : BIG, lz = valid[0] ? qz[0] : BIG;
          float hx = valid[0] ? qx[0] : -BIG, hy = valid[0] ? qy[0] : -BIG, hz = valid[0] ? qz[0] : -BIG, dummy = 0.0f;
          wave_min3_max4(lx, ly, lz, hx, hy, hz, dummy);
          const uint32_t gp = wave_greedy_point(ix, 0.5f * (lx + hx), 0.5f * (ly + hy), 0.5f * (lz + hz), topbox_s, ts);
          if (gp != NO_INDEX) {
            const float4 t = ix.pts[gp];  // wave-uniform address
            if (valid[0]) {
              const float d = l2_simple(qx[0], qy[0], qz[0], t.x, t.y, t.z);
              if (d < fast.best[0]) {
                fast.seed(0, d, gp);
                seed_pos[0] = gp;   // (the winner's original index is read again below unless this seed wins: t0 is stale)
                t0[0] = t;
              }
            }
            if (flags & 2) start_leaf = gp / LEAF;
            // (measured and dropped: every lane also evaluating the whole 64-point block around that leaf -- the second
            // launch of an alignment 0.895 -> 0.99 ms at 10M points, profiles/r05_lane_search_ab.txt: its cost is not its radii)
          }
        }
      }
    }
    // no lane has a seed: the first iteration of an alignment, queries stand off the target -> disc bounds
    ICP_LAP(5);
    traverse<NN1MinT<Q>, SPARSE>(ix, qx, qy, qz, valid, fast, wl_s[wave], topbox_s, ts, start_leaf, hm == 0);
    ICP_LAP(6);
    fast.resolve(ix, qx, qy, qz);
    if (!deep) {   // one group ahead: the next group's seed points, now that its seed positions are here
#pragma unroll
      for (int q = 0; q < Q; ++q)
        if (next_ok[q] && sp_n[q] != NO_INDEX) t_n[q] = ix.pts[sp_n[q]];
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      NN1 pol;
      pol.soa = ix.soa;
      pol.key = KEY_NONE;
      pol.pos = fast.bestpos[q];
      if (fast.bestpos[q] != NO_INDEX) {  // winner's original index: already here when the seed won
        const float w = (fast.bestpos[q] == seed_pos[q]) ? t0[q].w : ix.pts[fast.bestpos[q]].w;
        pol.key = make_key(fast.best[q], __float_as_uint(w));
      }
      const bool redo[1] = {valid[q] && (fast.tie[q] || (fast.bestpos[q] == NO_INDEX && !use_max))};
      if (__builtin_amdgcn_ballot_w64(redo[0]) != 0) {  // exact (distance, index) policy for tie lanes
        NN1 ex = pol;
        const float ex_x[1] = {qx[q]}, ex_y[1] = {qy[q]}, ex_z[1] = {qz[q]};
        traverse<NN1, SPARSE>(ix, ex_x, ex_y, ex_z, redo, ex, wl_s[wave], topbox_s, ts);
        if (redo[0]) pol = ex;
      }
      const uint32_t mid = key_index(pol.key);
      const bool found = valid[q] && mid != NO_INDEX;
      if (in_range[q]) {
        const uint32_t i = gcur * GROUP + q * WAVE + lane;
        match[i] = found ? mid : NO_INDEX;
        match_pos[i] = found ? pol.pos : NO_INDEX;
        match_d2[i] = found ? key_dist(pol.key) : __builtin_inff();
      }
    }
    if constexpr (OWNED) {
      // this launch's transform is in the group's working copy, its match entries are in use.  (Stored down here, behind
      // the traversal's cross-lane operations: every lane has read the old value by then in any execution order.)
      if (lane == 0) og.stamp[gcur] = epoch + 1u;
    }
    g = g2;
    if (deep) {
      gl_next = gl_nn;
      gl_nn = asked ? feed.resolve() : gl_after;
    } else {
      gl_next = asked ? feed.resolve() : gl_after;
    }
    ICP_LAP(7);
  }
#undef ICP_LAP
  flush_stats(ts, gstats);
}

// The launch WITHOUT seeds (first iteration of an alignment), for indices that carry leaf discs: a wave works through
// runs of consecutive -- spatially adjacent -- groups and seeds each from its predecessor's matches (standoff.hpp:
// collect / cull / evaluate); whatever that path gives up on goes through traverse() with the bounds reached so
// far.  Same outputs as the seeded search, bit for bit.
template <bool OWNED = false, class OG = NoOwnedGroups>
__device__ __forceinline__ void icp_cold_search_body(const IndexView& ix, float4* __restrict__ cur,
                                                     const float4* __restrict__ src0, uint32_t ns, Mat34 T,
                                                     const IcpControl* __restrict__ ctl, const RegionBox& region, int order,
                                                     float bound, int flags, float so_from,
                                                     uint32_t* __restrict__ match_pos, uint32_t* __restrict__ match,
                                                     float* __restrict__ match_d2, unsigned long long* gstats,
                                                     IcpWaveLds* wl_s, Box* topbox_s, const OG& og = OG()) {
  bool restart = false;
  if (ctl != nullptr) {
    if (ctl->stop != 0) return;
    restart = true;  // the device-driven loop only comes here for the first iteration of an alignment
#pragma unroll
    for (int i = 0; i < 12; ++i) T.m[i] = ctl->T_apply[i];
  }
  const float4* in = restart ? src0 : cur;
  const int use_max = flags & 1;
  load_top_cache(ix, topbox_s);
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = threadIdx.x / WAVE;
  const uint32_t GF = OWNED ? uint32_t(WAVE) : search_fill_of(flags);   // queries per wavefront (see icp_search_body)
  const bool lane_in_group = uint32_t(lane) < GF;
  uint32_t ngroups = (ns + GF - 1) / GF;
  if constexpr (OWNED) ngroups = og.count[0];  // slots of the served list (this body only starts alignments: no replay)
  const auto gid = [&](uint32_t slot) -> uint32_t {
    if constexpr (OWNED) return slot < ngroups ? og.list[slot] : 0u;
    else return slot;
  };
  TraverseStats ts;
  PrevGroup prev;
  prev.init();
  // Runs of COLD_RUN consecutive -- spatially adjacent -- groups of the XCD's window, handed out IN ORDER by the XCD's
  // counter (a wave's first run comes without asking; IndexView::sched_ctr, zeroed by PCLHIP_LAUNCH_FED): the waves of an
  // XCD work on one front that marches over the kd order, and whoever is through first takes the next run.  Inside a run
  // a group borrows its predecessor's match as its seed; a run starts without one (one lane's exact neighbour).  Until
  // round 3's last day every wave owned ONE run of ~38 groups (fixed shares: the launch ended with its slowest wave, and
  // the 4096 resident waves touched 4096 scattered neighbourhoods -- 2.3 GB fetched per launch): 2.58 -> 2.06 ms at 10M
  // points with runs of 4 (2.27 with 2: too many starts; 2.32 with 8: too coarse a tail).
  const GroupSchedule sched(ngroups);  // the XCD's window and this wave's slot in it
  constexpr uint32_t RUN = COLD_RUN;
  const uint32_t nruns = (sched.groups_per_xcd + RUN - 1u) / RUN;
  uint32_t* const run_ctr = ix.sched_ctr + (blockIdx.x % (gridDim.x < 8u ? gridDim.x : 8u)) * uint32_t(SCHED_CTR_STRIDE);
  for (uint32_t run = sched.slot_wave; run < nruns;) {
  uint32_t ticket = 0;
  if (lane == 0) ticket = atomicAdd(run_ctr, 1u);  // the run after this one: asked for now, read when this one is done
  prev.forget();
  uint32_t gl = run * RUN;
  const uint32_t run_end = (gl + RUN) < sched.groups_per_xcd ? (gl + RUN) : sched.groups_per_xcd;
  float4 p_n = make_float4(0, 0, 0, 0);
  {
    const uint32_t g0 = (gl < run_end) ? sched.global(gl) : ngroups;
    if (g0 < ngroups && lane_in_group && gid(g0) * GF + lane < ns) p_n = in[gid(g0) * GF + lane];
  }
  for (; gl < run_end; ++gl) {
    const uint32_t g = sched.global(gl);
    if (g >= ngroups) break;
    float4 p = p_n;
    const uint32_t i = gid(g) * GF + lane;
    const bool in_range = lane_in_group && i < ns;
    {  // the next group's points are in flight while this one is searched
      const uint32_t g2 = (gl + 1u < run_end) ? sched.global(gl + 1u) : ngroups;
      p_n = make_float4(0, 0, 0, 0);
      if (g2 < ngroups && lane_in_group && gid(g2) * GF + lane < ns) p_n = in[gid(g2) * GF + lane];
    }
    if constexpr (OWNED) {
      if (lane == 0) og.stamp[gid(g)] = 1u;  // the alignment's first transform applied; match entries in use
    }
    NN1Min fast;
    fast.init(bound);
    bool valid = in_range && isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
    if (valid) {
      const float x = xform_row(T.m[0], T.m[1], T.m[2], T.m[3], p.x, p.y, p.z, order);
      const float y = xform_row(T.m[4], T.m[5], T.m[6], T.m[7], p.x, p.y, p.z, order);
      const float z = xform_row(T.m[8], T.m[9], T.m[10], T.m[11], p.x, p.y, p.z, order);
      p.x = x; p.y = y; p.z = z;
      cur[i] = p;
      if (region.on) valid = in_region(region, x, y, z);  // target sharding: see icp_search_kernel
    } else if (in_range && restart) {
      cur[i] = p;  // non-finite points travel unchanged (icp.hpp:97-98)
    }
    const float qx[1] = {p.x}, qy[1] = {p.y}, qz[1] = {p.z};
    const bool vv[1] = {valid};
    uint32_t hint = NO_INDEX;
    const bool done =
        standoff_search(ix, p.x, p.y, p.z, valid, fast, wl_s[wave], topbox_s, ts, prev, (flags & 2) != 0, so_from, hint);
    prev.outcome(done);
#if defined(PCLHIP_SO_PROFILE) || defined(PCLHIP_SO_REASONS)
    const uint64_t so_t0 = clock64();
    TraverseStats ts_fb;
#else
    TraverseStats& ts_fb = ts;
#endif
    if (!done) traverse<NN1Min, true>(ix, qx, qy, qz, vv, fast, wl_s[wave], topbox_s, ts_fb, (flags & 2) ? hint : NO_INDEX, true);
    fast.resolve(ix, qx, qy, qz);
    NN1 pol;
    pol.soa = ix.soa;
    pol.key = KEY_NONE;
    pol.pos = fast.bestpos[0];
    if (fast.bestpos[0] != NO_INDEX) pol.key = make_key(fast.best[0], __float_as_uint(ix.pts[fast.bestpos[0]].w));
    {
      const bool redo[1] = {valid && (fast.tie[0] || (fast.bestpos[0] == NO_INDEX && !use_max))};
      if (__builtin_amdgcn_ballot_w64(redo[0]) != 0) {  // exact (distance, index) policy for tie lanes
        NN1 ex = pol;
        traverse<NN1, true>(ix, qx, qy, qz, redo, ex, wl_s[wave], topbox_s, ts_fb);
        if (redo[0]) pol = ex;
      }
    }
    const uint32_t mid = key_index(pol.key);
    const bool found = valid && mid != NO_INDEX;
    if (in_range) {
      match[i] = found ? mid : NO_INDEX;
      match_pos[i] = found ? pol.pos : NO_INDEX;
      match_d2[i] = found ? key_dist(pol.key) : __builtin_inff();
    }
    prev.record(p.x, p.y, p.z, found ? pol.pos : NO_INDEX);
#if defined(PCLHIP_SO_PROFILE)
    ts.c[7] += uint32_t(clock64() - so_t0);
    if (!done) ++ts.c[4];
#elif defined(PCLHIP_SO_REASONS)
    (void)so_t0;
    if (!done) ++ts.c[4];
#endif
  }
  run = sched.waves_per_xcd + uniform_u32(ticket);
  }
  flush_stats(ts, gstats);
}

// The kernels: the seeded search alone (host-driven iterations, fitness score), the stand-off search alone
// (host-driven first iteration), and both behind the control block of the device-driven loop -- one launch per
// iteration, IcpControl::restart picks the body on the device (registers and LDS are the maximum of the two bodies,
// which is what either needs anyway).
template <int MINW, int Q, bool SPARSE, bool DEEP = false>
__global__ __launch_bounds__(BLOCK, MINW) void icp_search_kernel(IndexView ix, float4* __restrict__ cur,
                                                                 const float4* __restrict__ src0, uint32_t ns,
                                                                 Mat34 T, const IcpControl* __restrict__ ctl,
                                                                 RegionBox region, int order, float bound, int flags,
                                                                 uint32_t* __restrict__ match_pos,
                                                                 uint32_t* __restrict__ match,
                                                                 float* __restrict__ match_d2,
                                                                 unsigned long long* gstats
                                                                 ) {
  __shared__ IcpWaveLds wl_s[WAVES_PER_BLOCK];
  __shared__ Box topbox_s[TOPCACHE_BOXES];
  icp_search_body<Q, SPARSE, false, NoOwnedGroups, DEEP>(ix, cur, src0, ns, T, ctl, region, order, bound, flags, match_pos, match,
                                                         match_d2, gstats, wl_s, topbox_s);
}

#ifndef PCLHIP_COLD_MINW
#define PCLHIP_COLD_MINW 4
#endif
__global__ __launch_bounds__(BLOCK, PCLHIP_COLD_MINW) void icp_cold_search_kernel(
    IndexView ix, float4* __restrict__ cur, const float4* __restrict__ src0, uint32_t ns, Mat34 T,
    const IcpControl* __restrict__ ctl, RegionBox region, int order, float bound, int flags, float so_from,
    uint32_t* __restrict__ match_pos, uint32_t* __restrict__ match, float* __restrict__ match_d2, unsigned long long* gstats) {
  __shared__ IcpWaveLds wl_s[WAVES_PER_BLOCK];
  __shared__ Box topbox_s[TOPCACHE_BOXES];
  icp_cold_search_body(ix, cur, src0, ns, T, ctl, region, order, bound, flags, so_from, match_pos, match, match_d2, gstats,
                       wl_s, topbox_s);
}

template <bool DEEP>
__global__ __launch_bounds__(BLOCK, PCLHIP_COLD_MINW) void icp_search_dual_kernel(
    IndexView ix, float4* __restrict__ cur, const float4* __restrict__ src0, uint32_t ns, Mat34 T,
    const IcpControl* __restrict__ ctl, RegionBox region, int order, float bound, int flags, float so_from,
    uint32_t* __restrict__ match_pos, uint32_t* __restrict__ match, float* __restrict__ match_d2, unsigned long long* gstats
    ) {
  __shared__ IcpWaveLds wl_s[WAVES_PER_BLOCK];
  __shared__ Box topbox_s[TOPCACHE_BOXES];
  if (ctl->restart != 0)
    icp_cold_search_body(ix, cur, src0, ns, T, ctl, region, order, bound, flags, so_from, match_pos, match, match_d2, gstats,
                         wl_s, topbox_s);
  else if ((flags & 4) != 0)
    return;  // SEARCH_RESTART_ONLY
  else
    icp_search_body<1, true, false, NoOwnedGroups, DEEP>(ix, cur, src0, ns, T, ctl, region, order, bound, flags, match_pos, match,
                                                         match_d2, gstats, wl_s, topbox_s);
}

// The same two bodies over the launch's list of SERVED groups (target sharding in the device-driven loop): `standoff`
// says whether launches that start an alignment take the stand-off body (the index carries discs and passes the gates).
__global__ __launch_bounds__(BLOCK, PCLHIP_COLD_MINW) void icp_search_owned_kernel(
    IndexView ix, float4* __restrict__ cur, const float4* __restrict__ src0, uint32_t ns, Mat34 T,
    const IcpControl* __restrict__ ctl, RegionBox region, int order, float bound, int flags, float so_from, int standoff,
    uint32_t* __restrict__ match_pos, uint32_t* __restrict__ match, float* __restrict__ match_d2, unsigned long long* gstats,
    OwnedGroups og) {
  __shared__ IcpWaveLds wl_s[WAVES_PER_BLOCK];
  __shared__ Box topbox_s[TOPCACHE_BOXES];
  if (standoff != 0 && ctl->restart != 0)
    icp_cold_search_body<true, OwnedGroups>(ix, cur, src0, ns, T, ctl, region, order, bound, flags, so_from, match_pos, match,
                                            match_d2, gstats, wl_s, topbox_s, og);
  else
    icp_search_body<1, true, true, OwnedGroups>(ix, cur, src0, ns, T, ctl, region, order, bound, flags, match_pos, match,
                                                match_d2, gstats, wl_s, topbox_s, og);
}

// ---- served groups (pclhip_internal.hpp: OwnedGroups) ----------------------------------------------------------------
// box of every 64-point group of the pristine, kd-ordered source: one wavefront per group
__global__ __launch_bounds__(BLOCK) void icp_group_box_kernel(const float4* __restrict__ src0, uint32_t ns,
                                                              float4* __restrict__ gbox) {
  const uint32_t g = blockIdx.x * WAVES_PER_BLOCK + threadIdx.x / WAVE;
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  const uint32_t ngroups = (ns + WAVE - 1) / WAVE;
  if (g >= ngroups) return;  // wave-uniform
  const float BIG = 3.402823466e+38f;
  float lx = BIG, ly = BIG, lz = BIG, hx = -BIG, hy = -BIG, hz = -BIG, unused = 0.0f;
  const uint32_t i = g * WAVE + lane;
  if (i < ns) {
    const float4 p = src0[i];
    if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
      lx = hx = p.x; ly = hy = p.y; lz = hz = p.z;
    }
  }
  wave_min3_max4(lx, ly, lz, hx, hy, hz, unused);
  if (lane == 0) {
    gbox[2 * g] = make_float4(lx, ly, lz, 0.0f);      // lo > hi: no finite point, never served
    gbox[2 * g + 1] = make_float4(hx, hy, hz, 0.0f);
  }
}

// The launch's place in the history of the running alignment, derived from the state the LAST launch left (every thread
// that needs it computes it the same way; icp_own_list_kernel writes it down for the search kernel afterwards).
struct OwnEpoch {
  uint32_t e;
  bool overflow;
};
__device__ __forceinline__ OwnEpoch own_epoch(const IcpControl* __restrict__ ctl, const OwnedState* __restrict__ st) {
  const bool restart = ctl->restart != 0;
  OwnEpoch o;
  o.e = restart ? 0u : st->epoch + 1u;
  // no room to remember this launch: nobody may skip it (or any later one)
  o.overflow = (!restart && st->overflow != 0u) || o.e >= uint32_t(OWN_HIST_CAP);
  return o;
}

// One thread per group: is it served in this launch?  After the launch every point sits at M * (pristine point), M =
// ctl->final_T (the guess for the launch that starts an alignment, Tk ... T1 guess later on); the group's box goes
// through M in float and is widened by far more than the working copy -- moved step by step, in float -- can deviate
// from that product (2e-4 of the coordinates' size against ~1e-7 per step).  A group that stops being served has its match
// entries emptied once (the accumulation and the host never see stale pairs).
__global__ __launch_bounds__(BLOCK) void icp_own_flag_kernel(const IcpControl* __restrict__ ctl,
                                                             const OwnedState* __restrict__ st,
                                                             const float4* __restrict__ gbox, RegionBox region,
                                                             uint32_t ngroups, uint32_t ns, uint32_t* __restrict__ stamp,
                                                             uint32_t* __restrict__ flags, uint32_t* __restrict__ block_count,
                                                             uint32_t* __restrict__ match,
                                                             uint32_t* __restrict__ match_pos, float* __restrict__ match_d2) {
  if (ctl->stop != 0) return;
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = g < ngroups;
  const bool restart = ctl->restart != 0;
  const bool overflow = own_epoch(ctl, st).overflow;
  uint32_t sv = (restart || !live) ? 0u : stamp[g];
  float4 lo = make_float4(1.0f, 0.0f, 0.0f, 0.0f), hi = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (live) {
    lo = gbox[2 * g];
    hi = gbox[2 * g + 1];
  }
  bool own = false;
  if (lo.x <= hi.x) {
    if (overflow) {
      own = true;
    } else {
      const float* M = ctl->final_T;
      const float BIG = 3.402823466e+38f;
      float bl[3] = {BIG, BIG, BIG}, bh[3] = {-BIG, -BIG, -BIG}, mag = 0.0f;
      for (int c = 0; c < 8; ++c) {
        const float x = (c & 1) ? hi.x : lo.x, y = (c & 2) ? hi.y : lo.y, z = (c & 4) ? hi.z : lo.z;
        for (int r = 0; r < 3; ++r) {
          const float v = M[4 * r] * x + M[4 * r + 1] * y + M[4 * r + 2] * z + M[4 * r + 3];
          bl[r] = fminf(bl[r], v);
          bh[r] = fmaxf(bh[r], v);
          mag = fmaxf(mag, fabsf(M[4 * r] * x) + fabsf(M[4 * r + 1] * y) + fabsf(M[4 * r + 2] * z) + fabsf(M[4 * r + 3]));
        }
      }
      const float eps = 2e-4f * mag + 1e-30f;
      own = true;
      for (int r = 0; r < 3; ++r) own = own && (bh[r] + eps >= region.lo[r]) && (bl[r] - eps < region.hi[r]);
      if (!(mag < BIG)) own = true;  // a transform that overflows: decide per point
    }
  }
  own = own && live;
  if (live) {
    flags[g] = own ? 1u : 0u;
    if (!own && (sv >> 31) == 0u) {
      for (uint32_t i = g * WAVE; i < ns && i < (g + 1u) * WAVE; ++i) {
        match[i] = NO_INDEX;
        match_pos[i] = NO_INDEX;
        match_d2[i] = __builtin_inff();
      }
      sv |= 0x80000000u;
    }
    stamp[g] = sv;
  }
  // served groups of this block of BLOCK groups (icp_own_list_kernel turns the counts into list positions)
  __shared__ uint32_t wcnt[WAVES_PER_BLOCK];
  const unsigned long long b = __builtin_amdgcn_ballot_w64(own);
  if ((threadIdx.x & (WAVE - 1)) == 0) wcnt[threadIdx.x / WAVE] = uint32_t(__builtin_popcountll(b));
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t c = 0;
    for (int w = 0; w < WAVES_PER_BLOCK; ++w) c += wcnt[w];
    block_count[blockIdx.x] = c;
  }
}

// When the loop hands control back to the host: every group's working copy through the last launch's transform, so that
// whatever reads or moves the working cloud next (a host-driven pclhip_icp_iterate) finds what the full pass leaves.
__global__ __launch_bounds__(BLOCK) void icp_own_catchup_kernel(float4* __restrict__ cur, const float4* __restrict__ src0,
                                                                uint32_t ns, int order, uint32_t* __restrict__ stamp,
                                                                const OwnedState* __restrict__ st) {
  const uint32_t g = blockIdx.x * WAVES_PER_BLOCK + threadIdx.x / WAVE;
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  const uint32_t ngroups = (ns + WAVE - 1) / WAVE;
  if (g >= ngroups || st->overflow != 0u) return;  // (after an overflow every group was served in every launch)
  const uint32_t raw = stamp[g], have = raw & 0x7FFFFFFFu, want = st->epoch + 1u;
  if (have >= want) return;
  const uint32_t i = g * WAVE + lane;
  if (i < ns) {
    float4 p = have == 0u ? src0[i] : cur[i];
    if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) own_replay(st, have, want, order, p);
    cur[i] = p;
  }
  __builtin_amdgcn_wave_barrier();  // every lane has read the stamp before lane 0 replaces it
  if (lane == 0) stamp[g] = want | (raw & 0x80000000u);
}

// The ascending list of the served groups, its length, and the launch's entry in the history -- one launch after the flags
// (until round 4: a one-thread kernel for the history, three scan launches and a scatter).  A workgroup owns OWN_CHUNK
// consecutive groups: its first list position is the sum of the flag kernel's block counts before it (a few thousand
// values at most), inside the chunk every thread owns 16 consecutive flags and a prefix sum over the threads orders them.
constexpr uint32_t OWN_CHUNK = 16u * BLOCK;
__global__ __launch_bounds__(BLOCK) void icp_own_list_kernel(const IcpControl* __restrict__ ctl, OwnedState* __restrict__ st,
                                                             const uint32_t* __restrict__ flags,
                                                             const uint32_t* __restrict__ block_count, uint32_t ngroups,
                                                             uint32_t* __restrict__ list, uint32_t* __restrict__ tot) {
  if (ctl->stop != 0) return;
  __shared__ uint32_t red[WAVES_PER_BLOCK];
  const uint32_t t = threadIdx.x, lane = t & (WAVE - 1), wave = t / WAVE;
  const uint32_t n_counts = (ngroups + BLOCK - 1) / BLOCK;
  uint32_t before = 0;
  {
    uint32_t lim = blockIdx.x * (OWN_CHUNK / BLOCK);
    if (lim > n_counts) lim = n_counts;
    for (uint32_t i = t; i < lim; i += BLOCK) before += block_count[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    if (lane == 0) red[wave] = before;
    __syncthreads();
    before = 0;
    for (int w = 0; w < WAVES_PER_BLOCK; ++w) before += red[w];
    __syncthreads();
  }
  const uint32_t g0 = blockIdx.x * OWN_CHUNK + 16u * t;
  uint32_t bits = 0;
#pragma unroll
  for (uint32_t j = 0; j < 16u; ++j)
    if (g0 + j < ngroups && flags[g0 + j] != 0u) bits |= 1u << j;
  const uint32_t cnt = uint32_t(__builtin_popcount(bits));
  uint32_t incl = cnt;
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    const uint32_t v = __shfl_up(incl, o);
    if (int(lane) >= o) incl += v;
  }
  if (lane == WAVE - 1) red[wave] = incl;
  __syncthreads();
  uint32_t waves_before = 0;
  for (uint32_t w = 0; w < wave; ++w) waves_before += red[w];
  uint32_t pos = before + waves_before + incl - cnt;
  for (uint32_t m = bits; m != 0u; m &= m - 1u) list[pos++] = g0 + uint32_t(__builtin_ctz(m));
  if (blockIdx.x == gridDim.x - 1) {
    if (t == BLOCK - 1) tot[0] = pos;  // the last thread of the last chunk ends the list
    if (t == 0) {                      // nobody reads the state in this launch: the history moves on here
      const bool restart = ctl->restart != 0;
      const OwnEpoch oe = own_epoch(ctl, st);
      st->epoch = oe.e;
      st->overflow = oe.overflow ? 1u : 0u;
      if (oe.e < uint32_t(OWN_HIST_CAP))
        for (int i = 0; i < 12; ++i) st->hist[oe.e][i] = ctl->T_apply[i];
      (void)restart;
    }
  }
}

// Reciprocal correspondences (registration/include/pcl/registration/impl/correspondence_estimation.hpp:247-270): the
// pair (source i, target j) survives only if source point i is the nearest neighbour of target point j among the source
// points, no farther than the maximum distance.  The query is the matched target point; source point i is a candidate
// with a known distance, so it seeds the search of the source index (an unseeded 1-NN of 10M queries costs 3.3 ms
// there, this one as much as a seeded ICP iteration).  Slots are the source's kd order: a wave's 64 queries are the
// matches of 64 neighbouring source points.  Exact (distance, index) order as everywhere: distance ties go through NN1.
__global__ __launch_bounds__(BLOCK, 4) void recip_search_kernel(IndexView sx, const float4* __restrict__ tgt_pts,
                                                                const uint32_t* __restrict__ match_pos,
                                                                const float4* __restrict__ cur, uint32_t n, float max_d2,
                                                                int use_max, uint8_t* __restrict__ keep,
                                                                unsigned long long* gstats) {
  __shared__ IcpWaveLds wl_s[WAVES_PER_BLOCK];
  __shared__ Box topbox_s[TOPCACHE_BOXES];
  load_top_cache(sx, topbox_s);
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = threadIdx.x / WAVE;
  const uint32_t ngroups = (n + WAVE - 1) / WAVE;
  const GroupSchedule sched(ngroups);
  TraverseStats ts;
  GroupFeed feed(sched, sx.sched_ctr);
  for (uint32_t gl = feed.first(sched); gl != GroupFeed::END; gl = feed.advance()) {
    const uint32_t g = sched.global(gl);
    if (g >= ngroups) break;
    feed.ahead(gl);
    const uint32_t i = g * WAVE + lane;
    bool valid = i < n && keep[i] != 0;
    float4 p = make_float4(0, 0, 0, 0);
    uint32_t seed_pos = NO_INDEX;
    float4 sp = make_float4(0, 0, 0, 0);
    if (valid) {
      p = tgt_pts[match_pos[i]];
      if (i < sx.n) {  // the index's positions ARE the slots (it borrows the working copy): the pair's own source point
        seed_pos = i;
        sp = sx.pts[seed_pos];
      }
    }
    const float qx[1] = {p.x}, qy[1] = {p.y}, qz[1] = {p.z};
    const bool vv[1] = {valid};
    NN1Min fast;
    fast.init(__builtin_inff());
    if (valid && seed_pos != NO_INDEX) fast.seed(0, l2_simple(p.x, p.y, p.z, sp.x, sp.y, sp.z), seed_pos);
    // The seed is a distance bound only: the source index is REFITTED to the moved cloud (refit_boxes), so after a
    // rotation its nodes are no longer the cells of a kd partition and the start-level shortcut of traverse() (which
    // needs disjoint cells) does not apply -- the descent starts at the root.
    traverse<NN1Min, true>(sx, qx, qy, qz, vv, fast, wl_s[wave], topbox_s, ts);
    fast.resolve(sx, qx, qy, qz);
    NN1 pol;
    pol.soa = sx.soa;
    pol.key = KEY_NONE;
    pol.pos = fast.bestpos[0];
    if (fast.bestpos[0] != NO_INDEX) {
      const float w = (fast.bestpos[0] == seed_pos) ? sp.w : sx.pts[fast.bestpos[0]].w;
      pol.key = make_key(fast.best[0], __float_as_uint(w));
    }
    const bool redo[1] = {valid && (fast.tie[0] || fast.bestpos[0] == NO_INDEX)};
    if (__builtin_amdgcn_ballot_w64(redo[0]) != 0) {  // exact (distance, index) policy for tie lanes
      NN1 ex = pol;
      traverse<NN1, true>(sx, qx, qy, qz, redo, ex, wl_s[wave], topbox_s, ts);
      if (redo[0]) pol = ex;
    }
    if (valid) {
      const uint32_t id = key_index(pol.key);
      const bool ok = id != NO_INDEX && id == __float_as_uint(cur[i].w) && !(use_max && key_dist(pol.key) > max_d2);
      if (!ok) keep[i] = 0;
    }
  }
  flush_stats(ts, gstats);
}

pclhip_status launch_recip_search(pclhip_index* src_ix, const float4* tgt_pts, const uint32_t* match_pos,
                                  const float4* cur, uint32_t n, float max_d2, bool use_max, uint8_t* keep) {
  pclhip_ctx* ctx = src_ix->ctx;
  if (n == 0) return PCLHIP_OK;
  const uint32_t ngroups = (n + WAVE - 1) / WAVE;
  const int grid = resident_blocks(ctx, recip_search_kernel, ngroups);
  PCLHIP_LAUNCH_FED(ctx, recip_search_kernel, dim3(grid), dim3(BLOCK), 0, ctx->stream, src_ix->view(), tgt_pts, match_pos,
                     cur, n, max_d2, use_max ? 1 : 0, keep, ctx->stats);
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  return PCLHIP_OK;
}

// Per-pair terms of the three transformation estimators, accumulated per lane in fp64 and reduced in a
// fixed order (wave shuffles, then the block's waves through LDS): deterministic for a given grid.
//   MODE 0  raw sums for umeyama (common/include/pcl/common/impl/eigen.hpp:696-712)
//   MODE 1  TransformationEstimationPointToPlaneLLS, impl/transformation_estimation_point_to_plane_lls.hpp:182-241
//           (float products in the reference's order, double sums)
//   MODE 2  TransformationEstimationSymmetricPointToPlaneLLS,
//           impl/transformation_estimation_symmetric_point_to_plane_lls.hpp:161-190: n = n1 +- n2,
//           v = [(p+q) x n ; n], ATA += v v^T, ATb += v ((q-p).n) -- float terms, double sums (the
//           reference sums in float in Eigen's internal order, which is not reproducible)
template <int MODE>
struct PairAcc {
  static constexpr int NACC = (MODE == PCLHIP_ICP_POINT_TO_POINT) ? 15 : 27;
  double acc[NACC];
  double sum_d2;
  uint32_t cnt, skipped;
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    sum_d2 = 0.0;
    cnt = 0;
    skipped = 0;
  }
  // p: source point (already moved), n1: its normal (MODE 2), t: matched target point, n: its normal
  __device__ __forceinline__ void add(const float4 p, const float4 n1, const float4 t, const float4 n, float d2,
                                      bool enforce_same_direction) {
    ++cnt;
    sum_d2 += double(d2);
    if constexpr (MODE == PCLHIP_ICP_POINT_TO_PLANE) {
      if (isfinite(n.x) && isfinite(n.y) && isfinite(n.z)) {
        const float sx = p.x, sy = p.y, sz = p.z;
        const float nx = n.x, ny = n.y, nz = n.z;
        const double a = double(__fsub_rn(__fmul_rn(nz, sy), __fmul_rn(ny, sz)));
        const double b = double(__fsub_rn(__fmul_rn(nx, sz), __fmul_rn(nz, sx)));
        const double c = double(__fsub_rn(__fmul_rn(ny, sx), __fmul_rn(nx, sy)));
        acc[0] += a * a;  acc[1] += a * b;  acc[2] += a * c;
        acc[3] += a * double(nx); acc[4] += a * double(ny); acc[5] += a * double(nz);
        acc[6] += b * b;  acc[7] += b * c;
        acc[8] += b * double(nx); acc[9] += b * double(ny); acc[10] += b * double(nz);
        acc[11] += c * c;
        acc[12] += c * double(nx); acc[13] += c * double(ny); acc[14] += c * double(nz);
        acc[15] += double(__fmul_rn(nx, nx)); acc[16] += double(__fmul_rn(nx, ny));
        acc[17] += double(__fmul_rn(nx, nz)); acc[18] += double(__fmul_rn(ny, ny));
        acc[19] += double(__fmul_rn(ny, nz)); acc[20] += double(__fmul_rn(nz, nz));
        // :235  nx*dx + ny*dy + nz*dz - nx*sx - ny*sy - nz*sz, float, left to right
        float df = __fmul_rn(nx, t.x);
        df = __fadd_rn(df, __fmul_rn(ny, t.y));
        df = __fadd_rn(df, __fmul_rn(nz, t.z));
        df = __fsub_rn(df, __fmul_rn(nx, sx));
        df = __fsub_rn(df, __fmul_rn(ny, sy));
        df = __fsub_rn(df, __fmul_rn(nz, sz));
        const double d = double(df);
        acc[21] += a * d; acc[22] += b * d; acc[23] += c * d;
        acc[24] += double(nx) * d; acc[25] += double(ny) * d; acc[26] += double(nz) * d;
      } else {
        ++skipped;
      }
    } else if constexpr (MODE == PCLHIP_ICP_SYMMETRIC) {
      float nx, ny, nz;
      const float dot12 = __fadd_rn(__fadd_rn(__fmul_rn(n1.x, n.x), __fmul_rn(n1.y, n.y)), __fmul_rn(n1.z, n.z));
      if (enforce_same_direction && !(dot12 >= 0.0f)) {  // :169-174
        nx = __fsub_rn(n1.x, n.x); ny = __fsub_rn(n1.y, n.y); nz = __fsub_rn(n1.z, n.z);
      } else {
        nx = __fadd_rn(n1.x, n.x); ny = __fadd_rn(n1.y, n.y); nz = __fadd_rn(n1.z, n.z);
      }
      const bool ok = isfinite(p.x) && isfinite(p.y) && isfinite(p.z) && isfinite(t.x) && isfinite(t.y) &&
                      isfinite(t.z) && isfinite(nx) && isfinite(ny) && isfinite(nz);  // :180-183
      if (ok) {
        const float sx = __fadd_rn(p.x, t.x), sy = __fadd_rn(p.y, t.y), sz = __fadd_rn(p.z, t.z);
        float v[6];
        v[0] = __fsub_rn(__fmul_rn(sy, nz), __fmul_rn(sz, ny));  // (p+q) x n
        v[1] = __fsub_rn(__fmul_rn(sz, nx), __fmul_rn(sx, nz));
        v[2] = __fsub_rn(__fmul_rn(sx, ny), __fmul_rn(sy, nx));
        v[3] = nx; v[4] = ny; v[5] = nz;
        const float dx = __fsub_rn(t.x, p.x), dy = __fsub_rn(t.y, p.y), dz = __fsub_rn(t.z, p.z);
        const float r = __fadd_rn(__fadd_rn(__fmul_rn(dx, nx), __fmul_rn(dy, ny)), __fmul_rn(dz, nz));
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
          for (int j = i; j < 6; ++j) acc[k++] += double(__fmul_rn(v[i], v[j]));  // upper triangle, row-major
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[21 + i] += double(__fmul_rn(v[i], r));
      } else {
        ++skipped;
      }
    } else {
      const double sx = p.x, sy = p.y, sz = p.z, tx = t.x, ty = t.y, tz = t.z;
      acc[0] += sx; acc[1] += sy; acc[2] += sz;
      acc[3] += tx; acc[4] += ty; acc[5] += tz;
      acc[6] += tx * sx; acc[7] += tx * sy; acc[8] += tx * sz;
      acc[9] += ty * sx; acc[10] += ty * sy; acc[11] += ty * sz;
      acc[12] += tz * sx; acc[13] += tz * sy; acc[14] += tz * sz;
    }
  }
  // block partial -> partials[blockIdx.x][NS]; red_s: [WAVES_PER_BLOCK][NS] doubles of LDS
  __device__ __forceinline__ void store_block(double (*red_s)[NS], double* __restrict__ partials) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const double s = wave_sum_d(acc[i]);
      if (lane == 0) red_s[wave][i] = s;
    }
    const double s0 = wave_sum_d(sum_d2), s1 = wave_sum_d(double(cnt)), s2 = wave_sum_d(double(skipped));
    if (lane == 0) {
      for (int i = NACC; i < NS; ++i) red_s[wave][i] = 0.0;
      red_s[wave][27] = s0;
      red_s[wave][28] = s1;
      red_s[wave][29] = s2;
    }
    __syncthreads();
    if (threadIdx.x < NS) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < WAVES_PER_BLOCK; ++w) s += red_s[w][threadIdx.x];
      partials[size_t(blockIdx.x) * NS + threadIdx.x] = s;
    }
  }
};

// so3 part of Transformer (common/include/pcl/common/impl/transforms.hpp:83-96): r0*a + (r1*b + r2*c)
__device__ __forceinline__ float rotate_row(float r0, float r1, float r2, float a, float b, float c) {
  return __fadd_rn(__fmul_rn(r0, a), __fadd_rn(__fmul_rn(r1, b), __fmul_rn(r2, c)));
}

template <int MODE>
__global__ __launch_bounds__(BLOCK) void icp_accumulate_kernel(IndexView ix, const float4* __restrict__ cur, uint32_t ns,
                                                               const uint32_t* __restrict__ match_pos,
                                                               const float* __restrict__ match_d2,
                                                               const uint8_t* __restrict__ keep, Mat34 T,
                                                               const IcpControl* __restrict__ ctl,
                                                               float4* __restrict__ src_nrm,
                                                               const float4* __restrict__ src_nrm0, int enforce,
                                                               double* __restrict__ partials) {
  __shared__ double red_s[WAVES_PER_BLOCK][NS];
  bool restart = false;
  if (ctl != nullptr) {
    if (ctl->stop != 0) return;
    restart = ctl->restart != 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) T.m[i] = ctl->T_apply[i];
  }
  PairAcc<MODE> pa;
  pa.init();
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t step = gridDim.x * blockDim.x;
#ifndef PCLHIP_ACC_PAIRS
#define PCLHIP_ACC_PAIRS 2  // A/B: 1 = one pair per trip
#endif
  if constexpr (MODE != PCLHIP_ICP_SYMMETRIC && PCLHIP_ACC_PAIRS == 2) {
    // Two pairs per trip: both match positions, then both gathers of (target point, normal), are in flight together --
    // the kernel is a chain of two dependent memory levels per pair and nothing else (113 -> 111 us at 10M pairs).
    // Pairs are still added in the order i, i + step, ...: the sums are the same, bit for bit.
    for (; uint64_t(i) + step < ns; i += 2u * step) {
      const uint32_t ia = i, ib = i + step;
      const uint32_t pos_a = match_pos[ia], pos_b = match_pos[ib];
      const bool ok_a = pos_a != NO_INDEX && (keep == nullptr || keep[ia]);
      const bool ok_b = pos_b != NO_INDEX && (keep == nullptr || keep[ib]);
      float4 pa4 = make_float4(0, 0, 0, 0), ta = pa4, na = pa4, pb4 = pa4, tb = pa4, nb = pa4;
      float da = 0.0f, db = 0.0f;
      if (ok_a) {
        pa4 = cur[ia];
        ta = ix.pts[pos_a];
        if constexpr (MODE != PCLHIP_ICP_POINT_TO_POINT) na = ix.nrm[pos_a];
        da = match_d2[ia];
      }
      if (ok_b) {
        pb4 = cur[ib];
        tb = ix.pts[pos_b];
        if constexpr (MODE != PCLHIP_ICP_POINT_TO_POINT) nb = ix.nrm[pos_b];
        db = match_d2[ib];
      }
      const float4 zero = make_float4(0, 0, 0, 0);
      if (ok_a) pa.add(pa4, zero, ta, na, da, enforce != 0);
      if (ok_b) pa.add(pb4, zero, tb, nb, db, enforce != 0);
    }
  }
  for (; i < ns; i += step) {
    float4 n1 = make_float4(0, 0, 0, 0);
    if constexpr (MODE == PCLHIP_ICP_SYMMETRIC) {
      // the source normals move with the cloud (transformPointCloudWithNormals, icp.hpp:49-111 override
      // of IterativeClosestPointWithNormals): rotate by this iteration's incremental transform
      const float4 m = restart ? src_nrm0[i] : src_nrm[i];
      n1.x = rotate_row(T.m[0], T.m[1], T.m[2], m.x, m.y, m.z);
      n1.y = rotate_row(T.m[4], T.m[5], T.m[6], m.x, m.y, m.z);
      n1.z = rotate_row(T.m[8], T.m[9], T.m[10], m.x, m.y, m.z);
      n1.w = m.w;
      src_nrm[i] = n1;
    }
    const uint32_t pos = match_pos[i];
    if (pos == NO_INDEX) continue;
    if (keep != nullptr && !keep[i]) continue;  // rejected by the reciprocal test / rejector chain
    const float4 p = cur[i];
    const float4 t = ix.pts[pos];
    float4 n = make_float4(0, 0, 0, 0);
    if constexpr (MODE != PCLHIP_ICP_POINT_TO_POINT) n = ix.nrm[pos];
    pa.add(p, n1, t, n, match_d2[i], enforce != 0);
  }
  pa.store_block(red_s, partials);
}

// The accumulation over the launch's SERVED groups (target sharding in the device-driven loop): slot s of the list's
// 64 * count slots is point 64 * list[s / 64] + s % 64.  Same per-pair arithmetic, a fixed order of its own.
template <int MODE>
__global__ __launch_bounds__(BLOCK) void icp_accumulate_owned_kernel(IndexView ix, const float4* __restrict__ cur, uint32_t ns,
                                                                     const uint32_t* __restrict__ match_pos,
                                                                     const float* __restrict__ match_d2,
                                                                     const uint8_t* __restrict__ keep,
                                                                     const IcpControl* __restrict__ ctl, int enforce,
                                                                     double* __restrict__ partials, OwnedGroups og) {
  static_assert(MODE != PCLHIP_ICP_SYMMETRIC, "the symmetric objective moves the source normals of every point: full pass");
  __shared__ double red_s[WAVES_PER_BLOCK][NS];
  if (ctl->stop != 0) return;
  PairAcc<MODE> pa;
  pa.init();
  const uint64_t slots = uint64_t(og.count[0]) * WAVE;
  const float4 zero = make_float4(0, 0, 0, 0);
  for (uint64_t s = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; s < slots; s += uint64_t(gridDim.x) * blockDim.x) {
    const uint32_t i = og.list[s / WAVE] * WAVE + uint32_t(s % WAVE);
    if (i >= ns) continue;
    const uint32_t pos = match_pos[i];
    if (pos == NO_INDEX) continue;
    if (keep != nullptr && !keep[i]) continue;
    const float4 p = cur[i];
    const float4 t = ix.pts[pos];
    float4 n = zero;
    if constexpr (MODE != PCLHIP_ICP_POINT_TO_POINT) n = ix.nrm[pos];
    pa.add(p, zero, t, n, match_d2[i], enforce != 0);
  }
  pa.store_block(red_s, partials);
}

// TransformationEstimation::estimateRigidTransformation(cloud_src, cloud_tgt) for n given pairs
// (registration/include/pcl/registration/transformation_estimation.h:71-115): pair i = (src[i], tgt[i]).
template <int MODE>
__global__ __launch_bounds__(BLOCK) void estimate_pairs_kernel(const float4* __restrict__ src,
                                                               const float4* __restrict__ src_nrm,
                                                               const float4* __restrict__ tgt,
                                                               const float4* __restrict__ tgt_nrm,
                                                               const float* __restrict__ weights, uint32_t n,
                                                               int enforce, double* __restrict__ partials) {
  __shared__ double red_s[WAVES_PER_BLOCK][NS];
  PairAcc<MODE> pa;
  pa.init();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 p = src[i], t = tgt[i];
    float4 n1 = make_float4(0, 0, 0, 0), n2 = make_float4(0, 0, 0, 0);
    if constexpr (MODE == PCLHIP_ICP_SYMMETRIC) n1 = src_nrm[i];
    if constexpr (MODE != PCLHIP_ICP_POINT_TO_POINT) n2 = tgt_nrm[i];
    if constexpr (MODE == PCLHIP_ICP_POINT_TO_PLANE) {
      // TransformationEstimationPointToPlaneLLSWeighted (impl/transformation_estimation_point_to_plane_lls_weighted.hpp
      // :227-229): the same sums with the target normal scaled by the pair's weight (float)
      if (weights != nullptr) {
        const float w = weights[i];
        n2.x = __fmul_rn(n2.x, w); n2.y = __fmul_rn(n2.y, w); n2.z = __fmul_rn(n2.z, w);
      }
    }
    const float dx = __fsub_rn(p.x, t.x), dy = __fsub_rn(p.y, t.y), dz = __fsub_rn(p.z, t.z);
    pa.add(p, n1, t, n2, __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)), enforce != 0);
  }
  pa.store_block(red_s, partials);
}

// partials[nblocks][NS] -> sums[NS]; fixed summation order (row r of 32 takes every 32nd block in four
// interleaved chains, then the 32 rows are added in order), so the result does not depend on scheduling.
// With `solve_ctl` the same launch also closes the iteration (icp_solve_step below): no record to exchange,
// one launch less.
__device__ __forceinline__ void icp_solve_step(IcpControl* __restrict__ ctl, const double* __restrict__ sums,
                                               IcpStepRecord* __restrict__ log);

// The thread that closes the iteration afterwards keeps the 6x6 system, the control block and the step record in
// registers (94 of the 128 a 1024-thread block leaves per lane): icp_solve_step is inlined here -- as a called function
// it was allocated on its own and went through 1 KB of scratch per call, 70 us for this launch instead of 27.
constexpr int FINALIZE_THREADS = 1024;
__global__ __launch_bounds__(FINALIZE_THREADS) void icp_finalize_kernel(const double* __restrict__ partials, int nblocks,
                                                                        double* __restrict__ sums,
                                                                        const IcpControl* __restrict__ ctl = nullptr,
                                                                        IcpControl* __restrict__ solve_ctl = nullptr,
                                                                        IcpStepRecord* __restrict__ log = nullptr) {
  if (ctl != nullptr && ctl->stop != 0) return;
  __shared__ double red[32][NS + 1];
  __shared__ double total[NS];
  const int t = threadIdx.x % NS;  // NS == 32
  for (int r = threadIdx.x / NS; r < 32; r += FINALIZE_THREADS / NS) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // four independent chains: the loads overlap
    int b = r;
    for (; b + 32 * 15 < nblocks; b += 32 * 16) {  // sixteen rows in flight per thread, summed in the order of the loop below
      double v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = partials[size_t(b + 32 * i) * NS + t];
#pragma unroll
      for (int i = 0; i < 16; i += 4) {
        s0 += v[i];
        s1 += v[i + 1];
        s2 += v[i + 2];
        s3 += v[i + 3];
      }
    }
    for (; b + 96 < nblocks; b += 128) {
      s0 += partials[size_t(b) * NS + t];
      s1 += partials[size_t(b + 32) * NS + t];
      s2 += partials[size_t(b + 64) * NS + t];
      s3 += partials[size_t(b + 96) * NS + t];
    }
    for (; b < nblocks; b += 32) s0 += partials[size_t(b) * NS + t];
    red[r][t] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  if (threadIdx.x < NS) {
    double a = 0.0;
#pragma unroll
    for (int i = 0; i < 32; ++i) a += red[i][t];
    sums[t] = a;
    total[t] = a;
  }
  if (solve_ctl != nullptr) {
    __syncthreads();
    if (threadIdx.x == 0) icp_solve_step(solve_ctl, total, log);
  }
}

// Closes an iteration on the device: closed form of the estimator on the (all-reduced) record, final = Tk *
// final, DefaultConvergenceCriteria, and the control words of the next launch (impl/icp.hpp:204-238).
// One thread: ~2k double operations, a few microseconds -- the point is that nothing leaves the GPU.
__device__ __forceinline__ void icp_solve_step(IcpControl* __restrict__ ctl, const double* __restrict__ sums,
                                               IcpStepRecord* __restrict__ log) {
  if (ctl->stop != 0) return;
  IcpControl c = *ctl;
  IcpStepRecord r;
  r.step = c.step;
  const double ncorr = sums[28];
  r.num_correspondences = ncorr;
  r.mse = 0.0;
  bool converged = false;
  if (ncorr < double(c.crit.min_number_correspondences)) {  // icp.hpp:204-213
    c.st.convergence_state = cf::NO_CORRESPONDENCES;
  } else {
    cf::solve(sums, c.mode, c.Tk);                     // :216-217
    cf::mat4_mul_f32(c.Tk, c.final_T, c.final_T);      // :223
    ++c.nr_iterations;
    const double mse = sums[27] / ncorr;               // calculateMSE, default_convergence_criteria.h:262-270
    r.mse = mse;
    converged = cf::has_converged(c.crit, c.st, c.nr_iterations, c.Tk, mse);
  }
  const bool ended = c.st.convergence_state != cf::NOT_CONVERGED;
  r.iteration = c.nr_iterations;
  r.convergence_state = c.st.convergence_state;
  r.converged = converged ? 1 : 0;
  r.ended = ended ? 1 : 0;
  r.similar = c.st.iterations_similar_transforms;
  r.prev_mse = c.st.prev_mse;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    r.Tk[i] = c.Tk[i];
    r.final_T[i] = c.final_T[i];
  }
  if (ended) {
    if (c.auto_restart) {  // the next launch starts the next alignment from the input cloud and the guess
      c.restart = 1;
      c.nr_iterations = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) c.final_T[i] = c.guess[i];
#pragma unroll
      for (int i = 0; i < 12; ++i) c.T_apply[i] = c.guess[i];
    } else {
      c.stop = 1;
    }
  } else {
    c.restart = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) c.T_apply[i] = c.Tk[i];  // :220, applied by the next search launch
  }
  ++c.step;
  *ctl = c;
  log[r.step % c.log_capacity] = r;
  __threadfence_system();
}

__global__ __launch_bounds__(64) void icp_solve_kernel(IcpControl* __restrict__ ctl, const double* __restrict__ sums,
                                 IcpStepRecord* __restrict__ log) {
  if (threadIdx.x == 0 && blockIdx.x == 0) icp_solve_step(ctl, sums, log);
}

// flags of the search kernels: 1 a finite maximum distance is set, 2 seeded descents may start below the root
constexpr int SEARCH_SKIP_FLAG = 2;  // seeded descents may start below the root (traverse(): start_leaf)
constexpr int SEARCH_RESEED = 8;  // a group whose seeds are all far gets one fresh seed from a greedy walk (icp_search_body)
constexpr int SEARCH_RESTART_ONLY = 4;  // device-driven loop with the per-lane search (lane.hip): this launch only serves the
                                        // iteration that STARTS an alignment and falls through otherwise

template <int MODE>
static void launch_accumulate(pclhip_icp* icp, const IndexView& v, int ga, const uint8_t* keep, const Mat34& M,
                              const IcpControl* ctl, hipStream_t s) {
  hipLaunchKernelGGL(icp_accumulate_kernel<MODE>, dim3(ga), dim3(BLOCK), 0, s, v, icp->src_cur, icp->n, icp->match_pos,
                     icp->match_d2, keep, M, ctl, icp->src_nrm_cur, icp->src_nrm_sorted0,
                     icp->enforce_same_direction_normals ? 1 : 0, icp->partials);
}

// Served-group lists (OwnedGroups): the arrays of a registration, made on first use; the boxes of the source's groups
static pclhip_status ensure_owned_groups(pclhip_icp* icp) {
  pclhip_ctx* ctx = icp->ctx;
  const uint32_t ngroups = (icp->n + WAVE - 1) / WAVE;
  if (icp->own_block != nullptr && icp->own_groups == ngroups) return PCLHIP_OK;
  if (icp->own_block) dev_free(ctx, icp->own_block);
  icp->own_block = nullptr;
  const size_t g = ngroups ? ngroups : 1;
  const size_t nb = (g + SC_BLOCK - 1) / SC_BLOCK;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += (bytes + 255) & ~size_t(255);
    return at;
  };
  const size_t o_box = take(2 * g * sizeof(float4)), o_stamp = take(g * 4), o_flags = take(g * 4), o_prefix = take(g * 4),
               o_list = take(g * 4), o_tot = take(16), o_part = take(nb * sizeof(uint2)), o_state = take(sizeof(OwnedState));
  char* base = nullptr;
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &base, off));
  icp->own_block = base;
  icp->own_groups = ngroups;
  icp->own_gbox = reinterpret_cast<float4*>(base + o_box);
  icp->own_stamp = reinterpret_cast<uint32_t*>(base + o_stamp);
  icp->own_flags = reinterpret_cast<uint32_t*>(base + o_flags);
  icp->own_prefix = reinterpret_cast<uint32_t*>(base + o_prefix);
  icp->own_list = reinterpret_cast<uint32_t*>(base + o_list);
  icp->own_tot = reinterpret_cast<uint32_t*>(base + o_tot);
  icp->own_partial = reinterpret_cast<uint2*>(base + o_part);
  icp->own_state = reinterpret_cast<OwnedState*>(base + o_state);
  PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(base + o_stamp, 0, off - o_stamp, ctx->stream));
  if (ngroups)
    hipLaunchKernelGGL(icp_group_box_kernel, dim3((ngroups + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), dim3(BLOCK), 0, ctx->stream,
                       icp->src_sorted0, icp->n, icp->own_gbox);
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  return PCLHIP_OK;
}

pclhip_status owned_groups_catch_up(pclhip_icp* icp, int mode) {
  if (icp->own_block == nullptr || icp->region.on == 0 || icp->n == 0) return PCLHIP_OK;
  pclhip_ctx* ctx = icp->ctx;
  const uint32_t ngroups = (icp->n + WAVE - 1) / WAVE;
  hipLaunchKernelGGL(icp_own_catchup_kernel, dim3((ngroups + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), dim3(BLOCK), 0, ctx->stream,
                     icp->src_cur, icp->src_sorted0, icp->n, mode == PCLHIP_ICP_POINT_TO_POINT ? 0 : 1, icp->own_stamp,
                     icp->own_state);
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  return PCLHIP_OK;
}

// A source that is SPARSE against the target index (a small scan against a large map, spread over it): 64 consecutive
// points of its kd order span a sizeable part of the target, and a wavefront walks the tree for the box of all of them --
// measured at 10M target points (round 6): 256 source points 34 ms, 4096 points 6.7 ms, 65,536 points 1.2 ms per iteration
// (0.07 / 0.08 / 0.43 ms with this).  Such a source gets FEWER points per wavefront -- 64 >> shift, a power of two: aligned
// runs of the kd order are cells -- so that a group spans about SPARSE_SRC_SPAN target points.
static int sparse_source_shift(const pclhip_icp* icp) {
  // the target points UNDER the source: the share of the target's bounding box the source's own box covers (a scan that
  // covers a patch of the map is dense there although it is small against the whole map)
  constexpr double SPARSE_SRC_SPAN = 1024.0;
  // over the target's two widest axes only: the third is the thickness of a surface, where the ratio of two noise
  // levels says nothing (and leaving an axis out errs towards fewer points per wavefront, the safe side)
  double share = 1.0, et[3];
  for (int d = 0; d < 3; ++d) et[d] = double(icp->target->bbox_hi[d]) - double(icp->target->bbox_lo[d]);
  const int thin = et[0] <= et[1] ? (et[0] <= et[2] ? 0 : 2) : (et[1] <= et[2] ? 1 : 2);
  for (int d = 0; d < 3; ++d) {
    const double es = double(icp->src_hi[d]) - double(icp->src_lo[d]);
    if (d != thin && et[d] > 0.0 && es >= 0.0 && es < et[d]) share *= es / et[d];
  }
  const double under = double(icp->target->n) * (share > 1e-9 ? share : 1e-9);
  const double fd = under > 0.0 ? SPARSE_SRC_SPAN * double(icp->n) / under : double(WAVE);
  const uint64_t f = fd >= double(WAVE) ? uint64_t(WAVE) : uint64_t(fd);
  int shift = 0;
  while (shift < 6 && (uint64_t(WAVE) >> shift) > (f < 1 ? 1 : f)) ++shift;
  return shift;
}

// One iteration.  ev == nullptr: the host-driven form (T by value, the caller reads the record back);
// ev != nullptr: the device-driven form -- transform / restart / stop come from icp->ctl, the iteration is
// closed by icp_solve_kernel, and ev[0..3] are recorded before the search, after it, after the accumulation
// and after the solve.
pclhip_status launch_icp_iterate(pclhip_icp* icp, const float T[16], float max_d2, bool use_max, int mode,
                                 hipEvent_t* ev) {
  pclhip_ctx* ctx = icp->ctx;
  hipStream_t s = ctx->stream;
  const IndexView v = icp->target->view();
  const bool device_loop = ev != nullptr;
  const IcpControl* ctl = device_loop ? icp->ctl : nullptr;
  Mat34 M;
  for (int i = 0; i < 12; ++i) M.m[i] = T ? T[i] : 0.0f;
  // clouds with normals move through transformPointCloudWithNormals (Transformer order), plain clouds
  // through Matrix4f * Vector4f (impl/icp.hpp:49-111)
  const int order = (mode == PCLHIP_ICP_POINT_TO_POINT) ? 0 : 1;
  // candidates must be <= max_d2 (a float): strict bound just above it; +inf when unbounded
  const float bound = use_max ? std::nextafterf(max_d2, __builtin_inff()) : __builtin_inff();
  uint32_t ngroups = (icp->n + WAVE - 1) / WAVE;
  const bool filters = icp->reciprocal || !icp->rejectors.empty();
  bool solved = false;
  if (icp->n > 0) {
    // pipeline depth of the seeded body (icp_search_body: DEEP): two groups ahead from PCLHIP_DEEP_FROM groups per wave on
    const auto deep_for = [&](int grid_blocks, uint32_t groups) {
      return uint64_t(groups) >= uint64_t(PCLHIP_DEEP_FROM) * uint64_t(grid_blocks) * uint64_t(WAVES_PER_BLOCK);
    };
    auto ks = icp_search_kernel<4, 1, true, false>;
    // launches without seeds go through the stand-off search when the index carries leaf discs: the host-driven loop
    // knows which launch that is (pclhip_icp_reset cleared the seeds); in the device-driven loop the control block
    // picks the body on the device (icp_search_dual_kernel)
    // The stand-off search culls by leaf discs: it pays where the leaves are THIN against their width (a surface sampled
    // well above its noise: thickness ratio ~0.1 at the bench's 10M points).  Where the noise is of the order of the point
    // spacing (ratio 0.3 at 100M points of the same surface) every query needs tens of leaves whatever the bound, the
    // lists outgrow the LDS, and the seeded search is the faster one (measured: 76 against 108 ms at 100M).
    // 0.3 since round 6 (option "standoff_thickness"): the same surface at 70M points (ratio between 0.2 and 0.3) starts an
    // alignment in 24.4 ms by the stand-off body against 31.8 by traverse(); at 100M points (0.3) the two are equal (54 / 55 ms);
    // the cube / layers / clusters families lie above 0.3 and are slower by the stand-off body (round 6: gate opened by option)
    const float SO_THICKNESS = ctx->opt_standoff_thickness;
    // ... and a gate on the index size, OPEN by default since round 6.  Round 4 measured the two bodies equal at 12M points
    // and traverse() ahead beyond (3.5 / 3.6, 4.7 / 4.5, 8.0 / 7.2 ms at 12M / 15M / 20M points) and closed the gate at 640 MB
    // of index; the stand-off body has since become 1.4x faster and the gate had gone stale -- round 6, the same launch with
    // the gate at 640 MB / open: 12M points 2.96 / 2.07 ms, 16M 4.29 / 2.76, 20M 5.89 / 3.66, 30M 10.0 / 6.7, 50M 19.2 / 14.6
    // (`ms_per_step` -12 ... -15 % there); at 100M points the thickness gate above decides (noise 1e-4 against a spacing of
    // 2e-4: ratio 0.3) and nothing changes.  The option "standoff_max_mb" remains.
    const size_t SO_MAX_INDEX_BYTES = size_t(ctx->opt_standoff_max_mb) << 20;   // 640 MB (option "standoff_max_mb")
    const bool standoff = v.disc != nullptr && icp->target->disc_thickness < SO_THICKNESS &&
                          size_t(icp->target->n_pad) * 56u <= SO_MAX_INDEX_BYTES;
    const bool cold = standoff && !device_loop && icp->seeds_cleared;
    const bool host_restart = !device_loop && icp->seeds_cleared;
    icp->seeds_cleared = false;
    int kflags = (use_max ? 1 : 0) | SEARCH_SKIP_FLAG | (ctx->opt_reseed != 0 ? SEARCH_RESEED : 0);
    // wave radii (squared) up to so_from stay with traverse(): the groups that sit on the surface
    const float so_from = icp->target->leaf_diag2;
    // target sharding in the device-driven loop: list the groups this rank serves in this launch, walk the list
    // (the reciprocal test searches the whole moved source: every group's working copy has to be current)
    const bool owned = device_loop && icp->region.on != 0 && mode != PCLHIP_ICP_SYMMETRIC && ctx->opt_served_groups != 0 &&
                       !icp->reciprocal;
    OwnedGroups og = {nullptr, nullptr, nullptr, nullptr};
    if (owned) {
      pclhip_status st = ensure_owned_groups(icp);
      if (st != PCLHIP_OK) return st;
      og.list = icp->own_list;
      og.count = icp->own_tot;
      og.stamp = icp->own_stamp;
      og.state = icp->own_state;
    }
    // Seeded launches one lane per query (lane.hip) wherever the target carries the structure for it: in the device-driven
    // loop the kernels of this file then only serve the launch that starts an alignment (SEARCH_RESTART_ONLY) and the
    // lane kernels every other one -- each falls through in the other's case, the control block decides on the device;
    // the host-driven loop knows which launch it is.
    const bool lane = !owned && lane_search_available(icp);
    const bool lane_now = lane && (device_loop || !host_restart);
    if (lane && device_loop) kflags |= SEARCH_RESTART_ONLY;
    // a source that is sparse against the target gets fewer points per wavefront (sparse_source_shift).  Not with the
    // served-group lists (their groups are the 64-point groups of the source) and not for the per-lane search.
    if (!owned) {
      kflags |= sparse_source_shift(icp) << 8;
      const uint32_t fill = search_fill_of(kflags);
      ngroups = (icp->n + fill - 1) / fill;
    }
    (void)hipEventRecord(device_loop ? ev[0] : icp->ev0, s);
    if (lane_now && !device_loop) {
      // nothing of this file
    } else if (owned) {
      hipLaunchKernelGGL(icp_own_flag_kernel, dim3((ngroups + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, ctl, icp->own_state,
                         icp->own_gbox, icp->region, ngroups, icp->n, icp->own_stamp, icp->own_flags, icp->own_prefix, icp->match,
                         icp->match_pos, icp->match_d2);
      hipLaunchKernelGGL(icp_own_list_kernel, dim3((ngroups + OWN_CHUNK - 1) / OWN_CHUNK), dim3(BLOCK), 0, s, ctl, icp->own_state,
                         icp->own_flags, icp->own_prefix, ngroups, icp->own_list, icp->own_tot);
      const int go = resident_blocks(ctx, icp_search_owned_kernel, ngroups);
      PCLHIP_LAUNCH_FED(ctx, icp_search_owned_kernel, dim3(go), dim3(BLOCK), 0, s, v, icp->src_cur, icp->src_sorted0, icp->n, M,
                         ctl, icp->region, order, bound, kflags, so_from, standoff ? 1 : 0, icp->match_pos, icp->match,
                         icp->match_d2, ctx->stats, og);
    } else if (standoff && device_loop) {
      const int gd = resident_blocks(ctx, icp_search_dual_kernel<false>, ngroups);
      if (deep_for(gd, ngroups)) {
        PCLHIP_LAUNCH_FED(ctx, icp_search_dual_kernel<true>, dim3(gd), dim3(BLOCK), 0, s, v, icp->src_cur, icp->src_sorted0, icp->n, M,
                           ctl, icp->region, order, bound, kflags, so_from, icp->match_pos, icp->match, icp->match_d2, ctx->stats);
      } else {
        PCLHIP_LAUNCH_FED(ctx, icp_search_dual_kernel<false>, dim3(gd), dim3(BLOCK), 0, s, v, icp->src_cur, icp->src_sorted0, icp->n, M,
                           ctl, icp->region, order, bound, kflags, so_from, icp->match_pos, icp->match, icp->match_d2, ctx->stats);
      }
    } else if (cold) {
      const int gc = resident_blocks(ctx, icp_cold_search_kernel, ngroups);
      PCLHIP_LAUNCH_FED(ctx, icp_cold_search_kernel, dim3(gc), dim3(BLOCK), 0, s, v, icp->src_cur, icp->src_sorted0, icp->n, M,
                         ctl, icp->region, order, bound, kflags, so_from, icp->match_pos, icp->match, icp->match_d2,
                         ctx->stats);
    } else {
      const int gk = resident_blocks(ctx, ks, ngroups);
      if (deep_for(gk, ngroups)) ks = icp_search_kernel<4, 1, true, true>;
      PCLHIP_LAUNCH_FED(ctx, ks, dim3(gk), dim3(BLOCK), 0, s, v, icp->src_cur, icp->src_sorted0,
                         icp->n, M, ctl, icp->region, order, bound, kflags, icp->match_pos, icp->match, icp->match_d2, ctx->stats);
    }
    if (lane_now) {
      const pclhip_status lst = launch_lane_search(icp, M.m, ctl, order, bound, use_max);
      if (lst != PCLHIP_OK) return lst;
    }
    (void)hipEventRecord(device_loop ? ev[1] : icp->ev_mid, s);
    icp->mid_recorded = true;
    const uint8_t* keep = nullptr;
    if (filters) {
      pclhip_status st = apply_correspondence_filters(icp, max_d2, use_max);
      if (st != PCLHIP_OK) return st;
      keep = icp->keep;
    }
    constexpr int acc_per_cu = 4;  // blocks of the streaming accumulate kernel per CU (rows the reduction reads)
    int ga = ctx->num_cus * acc_per_cu;
    if (ga > icp->grid_blocks) ga = icp->grid_blocks;
    {  // small clouds: no more blocks than give every thread ~4 points -- each block leaves a row of partial sums that the
       // single-workgroup reduction has to read (2048 rows cost it more than the sums of a 65k-point cloud)
      const int need = int((uint64_t(icp->n) + 1023u) / 1024u);
      if (ga > need) ga = need < 64 ? 64 : need;
    }
    if (owned && mode == PCLHIP_ICP_POINT_TO_PLANE)
      hipLaunchKernelGGL(icp_accumulate_owned_kernel<PCLHIP_ICP_POINT_TO_PLANE>, dim3(ga), dim3(BLOCK), 0, s, v, icp->src_cur,
                         icp->n, icp->match_pos, icp->match_d2, keep, ctl, icp->enforce_same_direction_normals ? 1 : 0,
                         icp->partials, og);
    else if (owned)
      hipLaunchKernelGGL(icp_accumulate_owned_kernel<PCLHIP_ICP_POINT_TO_POINT>, dim3(ga), dim3(BLOCK), 0, s, v, icp->src_cur,
                         icp->n, icp->match_pos, icp->match_d2, keep, ctl, icp->enforce_same_direction_normals ? 1 : 0,
                         icp->partials, og);
    else if (mode == PCLHIP_ICP_POINT_TO_PLANE)
      launch_accumulate<PCLHIP_ICP_POINT_TO_PLANE>(icp, v, ga, keep, M, ctl, s);
    else if (mode == PCLHIP_ICP_SYMMETRIC)
      launch_accumulate<PCLHIP_ICP_SYMMETRIC>(icp, v, ga, keep, M, ctl, s);
    else
      launch_accumulate<PCLHIP_ICP_POINT_TO_POINT>(icp, v, ga, keep, M, ctl, s);
    (void)hipEventRecord(device_loop ? ev[2] : icp->ev1, s);
    // (Folding this reduction into the accumulate kernel -- last block done -- was tried: the 2048 device-scope
    // atomics on one counter cost ~100 us across the 8 XCDs, five times this 20 us launch.)
    solved = device_loop && !icp_is_sharded(icp);  // no record to exchange: the reduction launch closes the iteration
    hipLaunchKernelGGL(icp_finalize_kernel, dim3(1), dim3(FINALIZE_THREADS), 0, s, icp->partials, ga, icp->sums_dev, ctl,
                       solved ? icp->ctl : static_cast<IcpControl*>(nullptr), icp->steps);
  } else {
    if (device_loop) {
      (void)hipEventRecord(ev[0], s);
      (void)hipEventRecord(ev[1], s);
    }
    if (filters) {  // an empty source share still takes part in the collectives of the chain's selection passes
      const pclhip_status es = apply_empty_shard_collectives(icp);
      if (es != PCLHIP_OK) return es;
    }
    PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(icp->sums_dev, 0, NS * sizeof(double), s));
    if (device_loop) (void)hipEventRecord(ev[2], s);
  }
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  // multi-GPU: the record is summed over the ranks on this stream, between the reduction and the solve
  pclhip_status st = allreduce_record(icp);
  if (st != PCLHIP_OK) return st;
  if (device_loop) {
    if (!solved) hipLaunchKernelGGL(icp_solve_kernel, dim3(1), dim3(64), 0, s, icp->ctl, icp->sums_dev, icp->steps);
    (void)hipEventRecord(ev[3], s);
    PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  }
  return PCLHIP_OK;
}

// ---- Registration::getFitnessScore (registration/include/pcl/registration/impl/registration.hpp:132-168)
// Mean of the 1-NN squared distances that are <= max_range (the reference compares the SQUARED distance
// with max_range as given, in double).  Per-block partial (sum, count) pairs, summed on the host in
// block order: deterministic.
__global__ __launch_bounds__(BLOCK) void fitness_partial_kernel(const float* __restrict__ d2, uint32_t n,
                                                                double max_range, double* __restrict__ partials) {
  __shared__ double red_s[WAVES_PER_BLOCK][2];
  double s = 0.0, c = 0.0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float d = d2[i];
    if (double(d) <= max_range) {  // +inf (no match / non-finite source point) never passes
      s += double(d);
      c += 1.0;
    }
  }
  s = wave_sum_d(s);
  c = wave_sum_d(c);
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
  if (lane == 0) {
    red_s[wave][0] = s;
    red_s[wave][1] = c;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    double a = 0.0;
#pragma unroll
    for (int w = 0; w < WAVES_PER_BLOCK; ++w) a += red_s[w][threadIdx.x];
    partials[size_t(blockIdx.x) * 2 + threadIdx.x] = a;
  }
}

pclhip_status launch_fitness_score(pclhip_icp* icp, const float T[16], double max_range, double* score,
                                   uint64_t* nr) {
  pclhip_ctx* ctx = icp->ctx;
  hipStream_t s = ctx->stream;
  *score = DBL_MAX;
  *nr = 0;
  if (icp->n == 0 && !icp_is_sharded(icp)) return PCLHIP_OK;
  double sum = 0.0, cnt = 0.0;
  if (icp->n > 0) {
  const IndexView v = icp->target->view();
  Mat34 M;
  for (int i = 0; i < 12; ++i) M.m[i] = T[i];
  const uint32_t n = icp->n;
  const int gr = ctx->num_cus * 4;
  // scratch: transformed copy of the source, seed/match positions, match ids, distances, partials
  const size_t o_pos = size_t(n) * sizeof(float4), o_id = o_pos + size_t(n) * 4, o_d2 = o_id + size_t(n) * 4;
  const size_t o_part = (o_d2 + size_t(n) * 4 + 15) & ~size_t(15);
  pclhip_status st = ensure_scratch(ctx, o_part + size_t(gr) * 2 * sizeof(double));
  if (st != PCLHIP_OK) return st;
  char* base = static_cast<char*>(ctx->scratch);
  float4* cur = reinterpret_cast<float4*>(base);
  uint32_t* pos = reinterpret_cast<uint32_t*>(base + o_pos);
  uint32_t* id = reinterpret_cast<uint32_t*>(base + o_id);
  float* d2 = reinterpret_cast<float*>(base + o_d2);
  double* part = reinterpret_cast<double*>(base + o_part);
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(cur, icp->src_sorted0, size_t(n) * sizeof(float4), hipMemcpyDeviceToDevice, s));
  // the last iteration's matches are valid upper bounds for any pose: use them as seeds
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(pos, icp->match_pos, size_t(n) * 4, hipMemcpyDeviceToDevice, s));
  const int fit_flags = sparse_source_shift(icp) << 8;   // (a sparse source: fewer points per wavefront, as in the iterations)
  const uint32_t fit_fill = search_fill_of(fit_flags);
  const uint32_t ngroups = (n + fit_fill - 1) / fit_fill;
  const int gs = resident_blocks(ctx, icp_search_kernel<4, 1, true>, ngroups);
  // transformPointCloud(cloud, out, Matrix4) is Transformer::se3 (transforms.hpp:109-123): order 1
  // target sharding: this rank scores the source points whose position under T lies in its region (every point has
  // exactly one owner), against its slab + halo index; the (sum, count) pairs are summed over the ranks below
  PCLHIP_LAUNCH_FED(ctx, (icp_search_kernel<4, 1, true>), dim3(gs), dim3(BLOCK), 0, s, v, cur, static_cast<const float4*>(cur), n,
                     M, static_cast<const IcpControl*>(nullptr), icp->region, 1, __builtin_inff(), fit_flags, pos, id, d2,
                     ctx->stats);
  hipLaunchKernelGGL(fitness_partial_kernel, dim3(gr), dim3(BLOCK), 0, s, d2, n, max_range, part);
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  std::vector<double> h(size_t(gr) * 2);
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(h.data(), part, h.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
  for (int b = 0; b < gr; ++b) {
    sum += h[size_t(b) * 2];
    cnt += h[size_t(b) * 2 + 1];
  }
  }  // (a rank whose share of the source is empty scores nothing but still takes part in the sum below)
  if (icp_is_sharded(icp)) {  // source slabs or target regions: one score for the whole registration
    std::memset(icp->sums_host, 0, NS * sizeof(double));
    icp->sums_host[0] = sum;
    icp->sums_host[1] = cnt;
    PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(icp->sums_dev, icp->sums_host, NS * sizeof(double), hipMemcpyHostToDevice, s));
    const pclhip_status ar = allreduce_record(icp);
    if (ar != PCLHIP_OK) return ar;
    PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(icp->sums_host, icp->sums_dev, NS * sizeof(double), hipMemcpyDeviceToHost, s));
    PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
    sum = icp->sums_host[0];
    cnt = icp->sums_host[1];
  }
  *nr = uint64_t(cnt);
  if (cnt > 0.0) *score = sum / cnt;
  return PCLHIP_OK;
}

// sums[NS] (host) of the estimator `mode` over n explicit pairs (dense float4 device arrays)
pclhip_status launch_estimate_pairs(pclhip_ctx* ctx, int mode, const float4* src, const float4* src_nrm,
                                    const float4* tgt, const float4* tgt_nrm, const float* weights, uint32_t n,
                                    bool enforce, double* sums) {
  hipStream_t s = ctx->stream;
  const int grid = ctx->num_cus * 4;
  double* dev = nullptr;  // partials [grid][NS] + sums [NS]
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &dev, (size_t(grid) + 1) * NS * sizeof(double)));
  if (mode == PCLHIP_ICP_POINT_TO_PLANE)
    hipLaunchKernelGGL(estimate_pairs_kernel<PCLHIP_ICP_POINT_TO_PLANE>, dim3(grid), dim3(BLOCK), 0, s, src, src_nrm, tgt,
                       tgt_nrm, weights, n, enforce ? 1 : 0, dev);
  else if (mode == PCLHIP_ICP_SYMMETRIC)
    hipLaunchKernelGGL(estimate_pairs_kernel<PCLHIP_ICP_SYMMETRIC>, dim3(grid), dim3(BLOCK), 0, s, src, src_nrm, tgt,
                       tgt_nrm, weights, n, enforce ? 1 : 0, dev);
  else
    hipLaunchKernelGGL(estimate_pairs_kernel<PCLHIP_ICP_POINT_TO_POINT>, dim3(grid), dim3(BLOCK), 0, s, src, src_nrm, tgt,
                       tgt_nrm, weights, n, enforce ? 1 : 0, dev);
  hipLaunchKernelGGL(icp_finalize_kernel, dim3(1), dim3(FINALIZE_THREADS), 0, s, dev, grid, dev + size_t(grid) * NS,
                     static_cast<const IcpControl*>(nullptr), static_cast<IcpControl*>(nullptr),
                     static_cast<IcpStepRecord*>(nullptr));
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(sums, dev + size_t(grid) * NS, NS * sizeof(double), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)dev_free(ctx, dev);
  PCLHIP_CHECK_HIP(ctx, e);
  return PCLHIP_OK;
}

// upper bound of the persistent grid (sizes the partial-sum buffer)
int icp_grid_blocks(pclhip_ctx* ctx, uint32_t ns) {
  const uint32_t ngroups = (ns + WAVE - 1) / WAVE;
  int g = resident_blocks(ctx, icp_search_kernel<4, 1, true>, ngroups);
  if (ctx->num_cus * 8 > g) g = ctx->num_cus * 8;  // the streaming accumulate kernel
  return g;
}


bool search_built_with_verify_bounds() {
#ifdef PCLHIP_VERIFY_BOUNDS
  return true;
#else
  return false;
#endif
}

// every kernel of this translation unit lives in one code object: asking for the occupancy of the hot ones loads it and
// fills resident_blocks()'s cache, so neither sits inside a first timed launch
void preload_search_kernels(pclhip_ctx* ctx) {
  (void)resident_blocks(ctx, icp_search_kernel<4, 1, true>, 1);
  (void)resident_blocks(ctx, icp_cold_search_kernel, 1);
  (void)resident_blocks(ctx, icp_search_dual_kernel<false>, 1);
  (void)resident_blocks(ctx, icp_search_dual_kernel<true>, 1);
  (void)resident_blocks(ctx, icp_search_kernel<4, 1, true, true>, 1);
  (void)resident_blocks(ctx, normals_kernel<8>, 1);
  (void)resident_blocks(ctx, knn_reg_kernel<1>, 1);
}

}  // namespace pclhip
