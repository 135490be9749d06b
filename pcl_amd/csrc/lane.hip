// lane.hip -- the SEEDED search launches of an ICP iteration, one lane per query (lane_search.hpp; round 5).
//
// Replaces, for every launch that is not the first of an alignment, the wave-cooperative icp_search_body of search.hip
// inside CorrespondenceEstimation::determineCorrespondences
// (registration/include/pcl/registration/impl/correspondence_estimation.hpp:145-218): same inputs (the working cloud,
// the previous matches as seeds), same outputs (match / match_pos / match_d2), bit for bit.  Three launches:
//
//   icp_lane_resolve_kernel   every source point: move it, evaluate its seed's leaf (a seed that is far away -- the
//                             second iteration of an alignment slides every query tens of point spacings along the
//                             surface -- is replaced by a greedy descent to a leaf next to the query), walk up at most
//                             `max_up` quad levels for a node whose cell holds the ball, search that node.  A lane whose
//                             ball does not fit such a node, or whose best distance is tied, gives up: its bit goes
//                             into the wave's mask.  (One lane that climbs to the root would hold 63 finished ones.)
//   icp_lane_queue_kernel     the given-up queries in source (= kd) order: masks -> ascending list, deterministic
//   icp_lane_finish_kernel    the list, 64 given-up queries per wave: the same search without the cap, under the exact
//                             (distance, index) policy NN1, so ties are settled here and nothing is left over.
//
// The launch that STARTS an alignment (queries stand off the target: the ball of every query spans hundreds of leaves)
// stays with the stand-off search of search.hip.
#include <hip/hip_runtime.h>

#include <cmath>

#include "icp_xform.hpp"
#include "lane_search.hpp"

namespace pclhip {

namespace {

constexpr int LBLOCK = 256;
constexpr int LWAVES = LBLOCK / WAVE;
constexpr uint32_t LQ_MASKS = 4;                      // masks (64 queries each) per thread of the queue kernel
constexpr uint32_t LQ_CHUNK = LQ_MASKS * LBLOCK;      // masks per workgroup = LQ_CHUNK / LWAVES resolve blocks
#ifndef PCLHIP_LANE_MINW
#define PCLHIP_LANE_MINW 6  // waves per SIMD the resolve kernel is compiled for (80 VGPRs, no spills; 8 = 64 VGPRs spills 99)
#endif

__device__ __forceinline__ void lane_stat(unsigned long long* g, int slot, bool flag) {
  const unsigned long long b = __builtin_amdgcn_ballot_w64(flag);
  if (g != nullptr && (threadIdx.x & (WAVE - 1)) == 0 && b != 0ull)
    atomicAdd(g + slot, static_cast<unsigned long long>(__builtin_popcountll(b)));
}

__global__ __launch_bounds__(LBLOCK, PCLHIP_LANE_MINW) void icp_lane_resolve_kernel(
    const float4* __restrict__ pts, const float* __restrict__ soa, LaneTree lt, float4* __restrict__ cur, uint32_t ns, Mat34 T,
    const IcpControl* __restrict__ ctl, RegionBox region, int order, float bound, int use_max, float far2, int max_up,
    uint32_t* __restrict__ match_pos, uint32_t* __restrict__ match, float* __restrict__ match_d2,
    unsigned long long* __restrict__ qmask, uint32_t* __restrict__ block_count, unsigned long long* gstats) {
  if (ctl != nullptr) {
    if (ctl->stop != 0 || ctl->restart != 0) return;  // ended / the launch that starts an alignment: not ours
#pragma unroll
    for (int i = 0; i < 12; ++i) T.m[i] = ctl->T_apply[i];
  }
  const uint32_t i = blockIdx.x * LBLOCK + threadIdx.x;
  const bool in_range = i < ns;
  float4 p = make_float4(0, 0, 0, 0);
  uint32_t sp = NO_INDEX;
  if (in_range) {
    p = cur[i];
    sp = match_pos[i];
  }
  bool valid = in_range && isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
  if (valid) {  // non-finite points travel unchanged (icp.hpp:97-98): nothing to write for them here
    const float x = xform_row(T.m[0], T.m[1], T.m[2], T.m[3], p.x, p.y, p.z, order);
    const float y = xform_row(T.m[4], T.m[5], T.m[6], T.m[7], p.x, p.y, p.z, order);
    const float z = xform_row(T.m[8], T.m[9], T.m[10], T.m[11], p.x, p.y, p.z, order);
    p.x = x; p.y = y; p.z = z;
    cur[i] = p;
    if (region.on) valid = in_region(region, x, y, z);  // target sharding: see icp_search_body
  }
  const float qx[1] = {p.x}, qy[1] = {p.y}, qz[1] = {p.z};
  NN1Min fast;
  fast.init(bound);
  bool gave_up = false, far = false, at_leaf = false;
  float4 t0 = make_float4(0, 0, 0, 0);
  if (valid) {
    far = sp == NO_INDEX;
    if (!far) {
      t0 = pts[sp];
      const float d = l2_simple(p.x, p.y, p.z, t0.x, t0.y, t0.z);
      fast.seed(0, d, sp);
      far = d > far2;
    }
    const uint32_t home = far ? lane_greedy_leaf(lt, p.x, p.y, p.z) : sp / LEAF;
    fast.leaf_global(soa, home, qx, qy, qz);
    if (gstats != nullptr) {  // (counted apart: how many are done with their own leaf)
      // (a one-leaf index stores no cell: the root's is all of space and is never written -- do not read it)
      at_leaf = lt.top == 0 || ball_inside_cell(lt.qcell[home], p.x, p.y, p.z, fast.best[0]);
    }
    gave_up = !lane_search(lt, soa, p.x, p.y, p.z, fast, home, max_up);
  }
  fast.resolve(soa, qx, qy, qz);
  uint32_t pos = NO_INDEX;
  if (valid) {
    // a tied distance, or an unbounded search that found nothing finite: the exact policy's business
    gave_up = gave_up || fast.tie[0] || (fast.bestpos[0] == NO_INDEX && !use_max);
    pos = fast.bestpos[0];
  }
  if (in_range) {
    if (!gave_up) {
      const bool found = valid && pos != NO_INDEX;
      uint32_t mid = NO_INDEX;
      if (found) mid = __float_as_uint(pos == sp ? t0.w : pts[pos].w);
      match[i] = found ? mid : NO_INDEX;
      match_pos[i] = found ? pos : NO_INDEX;
      match_d2[i] = found ? fast.best[0] : __builtin_inff();
    } else if (pos != NO_INDEX) {
      match_pos[i] = pos;  // the finishing pass starts from the best point seen here (any point is a valid bound)
    }
  }
  // the wave's mask of given-up queries, the block's count of them
  const unsigned long long mask = __builtin_amdgcn_ballot_w64(gave_up);
  __shared__ uint32_t wcnt[LWAVES];
  const uint32_t lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
  if (lane == 0) {
    const uint32_t g = blockIdx.x * LWAVES + wave;
    if (g * WAVE < ns) qmask[g] = mask;
    wcnt[wave] = uint32_t(__builtin_popcountll(mask));
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t c = 0;
    for (int w = 0; w < LWAVES; ++w) c += wcnt[w];
    block_count[blockIdx.x] = c;
  }
  if (gstats != nullptr) {  // pclhip_ctx_stats: 0 queries, 1 done with their own leaf, 2 done in this pass, 3 greedy descents
    lane_stat(gstats, 0, valid);
    lane_stat(gstats, 1, valid && at_leaf && !gave_up);
    lane_stat(gstats, 2, valid && !gave_up);
    lane_stat(gstats, 3, valid && far);
  }
}

// masks -> the ascending list of given-up queries.  A workgroup owns LQ_CHUNK masks: its first list position is the sum
// of the resolve kernel's block counts in front of it, inside the chunk a prefix sum over the threads orders the bits.
__global__ __launch_bounds__(LBLOCK) void icp_lane_queue_kernel(const IcpControl* __restrict__ ctl,
                                                                const unsigned long long* __restrict__ qmask,
                                                                const uint32_t* __restrict__ block_count, uint32_t ngroups,
                                                                uint32_t nblocks, uint32_t* __restrict__ queue,
                                                                uint32_t* __restrict__ tot) {
  if (ctl != nullptr && (ctl->stop != 0 || ctl->restart != 0)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) tot[0] = 0;
    return;
  }
  __shared__ uint32_t red[LWAVES];
  const uint32_t t = threadIdx.x, lane = t & (WAVE - 1), wave = t / WAVE;
  uint32_t before = 0;
  {
    uint32_t lim = blockIdx.x * (LQ_CHUNK / LWAVES);
    if (lim > nblocks) lim = nblocks;
    for (uint32_t i = t; i < lim; i += LBLOCK) before += block_count[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    if (lane == 0) red[wave] = before;
    __syncthreads();
    before = 0;
    for (int w = 0; w < LWAVES; ++w) before += red[w];
    __syncthreads();
  }
  const uint32_t g0 = blockIdx.x * LQ_CHUNK + LQ_MASKS * t;
  unsigned long long m[LQ_MASKS];
  uint32_t cnt = 0;
#pragma unroll
  for (uint32_t j = 0; j < LQ_MASKS; ++j) {
    m[j] = (g0 + j < ngroups) ? qmask[g0 + j] : 0ull;
    cnt += uint32_t(__builtin_popcountll(m[j]));
  }
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
#pragma unroll
  for (uint32_t j = 0; j < LQ_MASKS; ++j)
    for (unsigned long long b = m[j]; b != 0ull; b &= b - 1ull) queue[pos++] = (g0 + j) * WAVE + uint32_t(__builtin_ctzll(b));
  if (blockIdx.x == gridDim.x - 1 && t == LBLOCK - 1) tot[0] = pos;  // the last thread of the last chunk ends the list
}

__global__ __launch_bounds__(LBLOCK) void icp_lane_finish_kernel(const float4* __restrict__ pts, const float* __restrict__ soa,
                                                                 LaneTree lt, const float4* __restrict__ cur,
                                                                 const IcpControl* __restrict__ ctl, float bound, int use_max,
                                                                 float far2, uint32_t* __restrict__ match_pos,
                                                                 uint32_t* __restrict__ match, float* __restrict__ match_d2,
                                                                 const uint32_t* __restrict__ queue,
                                                                 const uint32_t* __restrict__ tot, unsigned long long* gstats) {
  if (ctl != nullptr && (ctl->stop != 0 || ctl->restart != 0)) return;
  const uint32_t count = tot[0];
  uint32_t served = 0;
  for (uint32_t s = blockIdx.x * LBLOCK + threadIdx.x; s < count; s += gridDim.x * LBLOCK) {
    const uint32_t i = queue[s];
    const float4 p = cur[i];  // moved by the resolve kernel, finite, inside the rank's region
    const uint32_t sp = match_pos[i];
    const float qx[1] = {p.x}, qy[1] = {p.y}, qz[1] = {p.z};
    NN1 pol;
    pol.soa = soa;
    // candidates must be < bound (correspondence_estimation.hpp:186: distances beyond the maximum are dropped); an
    // unbounded search takes whatever is nearest, a point at an overflowing distance included
    pol.key = use_max ? make_key(bound, 0u) : KEY_NONE;
    pol.pos = NO_INDEX;
    bool far = sp == NO_INDEX;
    if (!far) {
      const float4 t0 = pts[sp];
      const float d = l2_simple(p.x, p.y, p.z, t0.x, t0.y, t0.z);
      const uint64_t k = make_key(d, __float_as_uint(t0.w));
      if (k < pol.key) {
        pol.key = k;
        pol.pos = sp;
      }
      far = d > far2;
    }
    const uint32_t home = far ? lane_greedy_leaf(lt, p.x, p.y, p.z) : sp / LEAF;
    pol.leaf_global(soa, home, qx, qy, qz);
    (void)lane_search(lt, soa, p.x, p.y, p.z, pol, home, 64);
    const bool found = pol.pos != NO_INDEX;
    match[i] = found ? key_index(pol.key) : NO_INDEX;
    match_pos[i] = found ? pol.pos : NO_INDEX;
    match_d2[i] = found ? key_dist(pol.key) : __builtin_inff();
    ++served;
  }
  if (gstats != nullptr && served != 0u) atomicAdd(gstats + 4, static_cast<unsigned long long>(served));  // 4: finished here
}

}  // namespace

bool lane_search_available(const pclhip_icp* icp) {
  return icp->target != nullptr && icp->target->qcell != nullptr && icp->ctx->opt_lane_search != 0;
}

static pclhip_status ensure_lane_buffers(pclhip_icp* icp) {
  pclhip_ctx* ctx = icp->ctx;
  if (icp->lane_block != nullptr && icp->lane_cap == icp->n) return PCLHIP_OK;
  if (icp->lane_block) dev_free(ctx, icp->lane_block);
  icp->lane_block = nullptr;
  const size_t n = icp->n ? icp->n : 1;
  const size_t ngroups = (n + WAVE - 1) / WAVE, nblocks = (n + LBLOCK - 1) / LBLOCK;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += (bytes + 255) & ~size_t(255);
    return at;
  };
  const size_t o_mask = take(ngroups * 8), o_cnt = take(nblocks * 4), o_queue = take(n * 4), o_tot = take(16);
  char* base = nullptr;
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &base, off));
  icp->lane_block = base;
  icp->lane_cap = icp->n;
  icp->lane_mask = reinterpret_cast<unsigned long long*>(base + o_mask);
  icp->lane_bcount = reinterpret_cast<uint32_t*>(base + o_cnt);
  icp->lane_queue = reinterpret_cast<uint32_t*>(base + o_queue);
  icp->lane_tot = reinterpret_cast<uint32_t*>(base + o_tot);
  return PCLHIP_OK;
}

// The three launches, stream-ordered.  With `ctl` (device-driven loop) they fall through when the control block says
// "stop" or "this launch starts an alignment".
pclhip_status launch_lane_search(pclhip_icp* icp, const float T12[12], const IcpControl* ctl, int order, float bound,
                                 bool use_max) {
  pclhip_ctx* ctx = icp->ctx;
  hipStream_t s = ctx->stream;
  if (icp->n == 0) return PCLHIP_OK;
  pclhip_status st = ensure_lane_buffers(icp);
  if (st != PCLHIP_OK) return st;
  const pclhip_index* ix = icp->target;
  const LaneTree lt = ix->lane_tree();
  Mat34 M;
  for (int i = 0; i < 12; ++i) M.m[i] = T12 ? T12[i] : 0.0f;
  const uint32_t ngroups = (icp->n + WAVE - 1) / WAVE, nblocks = (icp->n + LBLOCK - 1) / LBLOCK;
  // a seed farther than this (squared) is no seed: the lane looks for a leaf next to the query instead
  const float far2 = ctx->opt_lane_far * ix->leaf_diag2;
  hipLaunchKernelGGL(icp_lane_resolve_kernel, dim3(nblocks), dim3(LBLOCK), 0, s, ix->pts, ix->soa, lt, icp->src_cur, icp->n, M,
                     ctl, icp->region, order, bound, use_max ? 1 : 0, far2, ctx->opt_lane_max_up, icp->match_pos, icp->match,
                     icp->match_d2, icp->lane_mask, icp->lane_bcount, ctx->stats);
  hipLaunchKernelGGL(icp_lane_queue_kernel, dim3((ngroups + LQ_CHUNK - 1) / LQ_CHUNK), dim3(LBLOCK), 0, s, ctl, icp->lane_mask,
                     icp->lane_bcount, ngroups, nblocks, icp->lane_queue, icp->lane_tot);
  uint32_t gf = uint32_t(ctx->num_cus) * 8u;
  if (gf > nblocks) gf = nblocks;
  hipLaunchKernelGGL(icp_lane_finish_kernel, dim3(gf), dim3(LBLOCK), 0, s, ix->pts, ix->soa, lt, icp->src_cur, ctl, bound,
                     use_max ? 1 : 0, far2, icp->match_pos, icp->match, icp->match_d2, icp->lane_queue, icp->lane_tot,
                     ctx->stats);
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  return PCLHIP_OK;
}

void preload_lane_kernels() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(icp_lane_resolve_kernel));
}

}  // namespace pclhip
