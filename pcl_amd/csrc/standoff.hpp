// standoff.hpp -- exact 1-NN of a 64-query group that STANDS OFF the indexed surface (the unseeded first
// iteration of a registration: the queries sit tens of point spacings away from the target sheet).
//
// traverse() finds the leaves a group needs while it walks: every listed leaf is tested by all 64 lanes, the radii
// tighten on the way, and the work is set by the ~70 leaves the GROUP-level box bound cannot exclude (57 per-lane
// disc tests and 15 evaluation rounds per group at the bench's stand-off, for ~4 leaves a lane really needs).
// This path separates the steps instead (collect -> cull -> evaluate), which works because every lane STARTS with a
// near-tight radius:
//   0. seed      one target point near the group (the match of the previous group of this wave -- a wave owns
//                consecutive, i.e. spatially adjacent, groups in a cold launch -- or the exact neighbour of one lane
//                found with traverse()); every lane's radius is its distance to that point: within a few percent
//                of final, because the stand-off dominates the distance.
//   1. collect   one walk of the tree with the (fixed) wave radius and the exact box bound: no ordering, no
//                ranking; the surviving leaf ids go into one list in LDS (~100 entries).
//   2. row seeds every row of 16 lanes (a 4x4-spacing patch of the kd-ordered group) evaluates the two listed leaves
//                whose disc centres are nearest to the row's centre: radii within ~0.5 % of final.
//   3. row cull  (row, leaf) pairs by the reach filter (traverse.hpp: row_reach_alive), 64 pairs per pass; leaves alive
//                for any row are compacted into <= 64 union slots (disc + id) in LDS.
//   4. lane cull every lane runs the disc bound (point_disc_lb) against the slots alive for its row only.
//   5. evaluate  the needed leaves, lane-sparse, 16 staged leaves at a time (LDS-DMA, transposed).
// Exactness: a leaf that holds a point within a lane's current bound passes the box test against the wave radius
// (bit-monotone), the reach filter of the lane's row (row maxima bound the lane's own quantities) and the lane's disc
// test; bounds only shrink, so a stale (larger) one is conservative.  Both inexact bounds are the ones traverse()
// already uses (tests/test_reach_bound_model.py).  Anything that does not fit (list or union overflow, no finite
// radius) returns false and the caller runs traverse() with the policy state reached so far, which is always valid.
//
// LDS: the wave's WaveLdsT block is re-used stage by stage -- collect: stack[] + ids in buf[]; row seeds: staging in
// list[]; cull: union slots in list[] / rad[]; evaluate: staging in buf[] (the ids are dead by then).
#pragma once

#include <cstddef>

#include "traverse.hpp"

namespace pclhip {

// -DPCLHIP_SO_PROFILE (scripts/build_variant.sh): the work counters become clock64() ticks per stage --
// [0] seed, [1] collect, [2] group cull, [3] row seeds, [5] row cull, [6] lane cull, [7] evaluation + the rest of the
// kernel's group loop (fallback, resolve, ties, stores)
// -DPCLHIP_SO_REASONS: the counters become the number of groups that left the path at each exit --
// [0] no finite radius, [1] radius below `from2`, [2] list overflow, [3] frontier overflow, [5] disc overflow,
// [6] union overflow, [7] done
#ifdef PCLHIP_SO_REASONS
#define SO_WHY(i) ++ts.c[i]
#else
#define SO_WHY(i) (void)0
#endif
#ifdef PCLHIP_SO_MARK
#define SO_MARK(name) asm volatile("; SO_MARK " name)
#else
#define SO_MARK(name) (void)0
#endif
#if defined(PCLHIP_SO_PROFILE)
#define SO_LAP(i)                          \
  do {                                     \
    const uint64_t so_now = clock64();     \
    ts.c[i] += uint32_t(so_now - so_t);    \
    so_t = so_now;                         \
  } while (0)
#define SO_COUNT(expr) (void)0
#elif defined(PCLHIP_SO_STATS)  // per-step work counters ([0] nodes, [1] lane-cull steps, [2] evaluation rounds, [3] pushes)
#define SO_LAP(i) (void)0
#define SO_COUNT(expr) expr
#else  // the default build only counts per group: [4] groups, [5] groups finished here, [6] collected leaves, [7] union slots
#define SO_LAP(i) (void)0
#define SO_COUNT(expr) (void)0
#endif

#ifndef PCLHIP_SO_BATCHCULL
#define PCLHIP_SO_BATCHCULL 1  // A/B: 0 culls all union slots of a segment before the first evaluation
#endif

constexpr uint32_t SO_LIST_CAP = 768;   // collected leaf ids (they fill the 3 KB staging area)
constexpr uint32_t SO_DISC_CAP = 124;   // disc entries (32 B) in LDS at a time: bytes [0, 3968) of the wave's block
constexpr uint32_t SO_SURV_CAP = 192;   // ids of the leaves that pass the group cull: bytes [4224, 4992)
constexpr uint32_t SO_UNION_CAP = 64;   // union slots: one bit each in the per-row / per-lane masks

// DPP row operations through the builtin (the compiler places the wait states); used once or twice per group
template <int CTRL>
__device__ __forceinline__ uint32_t so_dpp(uint32_t v) {
  return uint32_t(__builtin_amdgcn_update_dpp(int(v), int(v), CTRL, 0xF, 0xF, false));
}
// every lane ends up with the minimum / the OR over its row of 16 lanes
__device__ __forceinline__ uint32_t row_min_u32(uint32_t v) {
  v = min(v, so_dpp<0xB1>(v));    // quad_perm [1,0,3,2]
  v = min(v, so_dpp<0x4E>(v));    // quad_perm [2,3,0,1]
  v = min(v, so_dpp<0x141>(v));   // row_half_mirror
  v = min(v, so_dpp<0x140>(v));   // row_mirror
  return v;
}
__device__ __forceinline__ uint32_t row_or_u32(uint32_t v) {
  v |= so_dpp<0xB1>(v);
  v |= so_dpp<0x4E>(v);
  v |= so_dpp<0x141>(v);
  v |= so_dpp<0x140>(v);
  return v;
}

// Geometry of the four rows of a group: centre and radius of every row's queries (rounded up), and the patch
// direction ng (the rows are the quadrants of the kd-ordered 64-query patch: the cross product of the two
// diagonals of their centres is its normal; any direction is valid, a poor one only filters less).
struct RowGeom {
  float cx, cy, cz, rS;
  float ngx, ngy, ngz;
  bool any;  // the row has a valid query
};
__device__ __forceinline__ RowGeom row_geometry(float qx, float qy, float qz, bool valid) {
  const float BIGF = 3.402823466e+38f;
  RowGeom g;
  float l0 = valid ? qx : BIGF, l1 = valid ? qy : BIGF, l2 = valid ? qz : BIGF;
  float h0 = valid ? qx : -BIGF, h1 = valid ? qy : -BIGF, h2 = valid ? qz : -BIGF;
  row_min3_f(l0, l1, l2);
  row_max3_f(h0, h1, h2);
  g.any = !(l0 > h0);
  g.cx = g.any ? 0.5f * (l0 + h0) : 0.0f;
  g.cy = g.any ? 0.5f * (l1 + h1) : 0.0f;
  g.cz = g.any ? 0.5f * (l2 + h2) : 0.0f;
  const float ex = g.any ? h0 - l0 : 0.0f, ey = g.any ? h1 - l1 : 0.0f, ez = g.any ? h2 - l2 : 0.0f;
  g.rS = __fsqrt_rn((ex * ex + ey * ey) + ez * ez) * 0.5000005f + 1e-6f * ((fabsf(g.cx) + fabsf(g.cy)) + fabsf(g.cz));
  const float r0x = readlane_f(g.cx, 0), r0y = readlane_f(g.cy, 0), r0z = readlane_f(g.cz, 0);
  const float r1x = readlane_f(g.cx, 16), r1y = readlane_f(g.cy, 16), r1z = readlane_f(g.cz, 16);
  const float r2x = readlane_f(g.cx, 32), r2y = readlane_f(g.cy, 32), r2z = readlane_f(g.cz, 32);
  const float r3x = readlane_f(g.cx, 48), r3y = readlane_f(g.cy, 48), r3z = readlane_f(g.cz, 48);
  const float ux = r3x - r0x, uy = r3y - r0y, uz = r3z - r0z, vx = r2x - r1x, vy = r2y - r1y, vz = r2z - r1z;
  const float cxn = uy * vz - uz * vy, cyn = uz * vx - ux * vz, czn = ux * vy - uy * vx;
  const float l2n = cxn * cxn + cyn * cyn + czn * czn;
  g.ngx = 0.0f;
  g.ngy = 0.0f;
  g.ngz = 0.0f;
  if (l2n > 1e-30f && l2n < 1e30f) {
    const float il = __frsqrt_rn(l2n);
    g.ngx = cxn * il;
    g.ngy = cyn * il;
    g.ngz = czn * il;
  }
  return g;
}
// the row maxima the reach filter needs, from the lanes' CURRENT bounds (squared distances)
__device__ __forceinline__ RowReach row_reach_of(const RowGeom& g, float qx, float qy, float qz, bool valid, float worst2) {
  const float BIGF = 3.402823466e+38f;
  RowReach rr;
  rr.cx = g.cx;
  rr.cy = g.cy;
  rr.cz = g.cz;
  rr.rS = g.rS;
  const float a_own = (g.ngx * (qx - g.cx) + g.ngy * (qy - g.cy)) + g.ngz * (qz - g.cz);
  const float rho = __fsqrt_rn(valid ? worst2 : 0.0f) * 1.000004f;
  rr.Up = valid ? rho - a_own : -BIGF;
  rr.Um = valid ? rho + a_own : -BIGF;
  rr.rho = valid ? rho : 0.0f;
  row_max3_f(rr.Up, rr.Um, rr.rho);
  return rr;
}

// v_sqrt_f32 as it is (1 ulp, denormal results flushed towards zero): every use below rounds its result towards
// the conservative side by a relative 1e-6 (sixteen times the error) or more, as the bounds in traverse.hpp do after
// their correctly rounded square roots.
__device__ __forceinline__ float so_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// row_reach_alive (traverse.hpp) with so_sqrt: same bound, same allowances
__device__ __forceinline__ bool so_reach_alive(const RowReach& rr, float ngx, float ngy, float ngz, const float4 cR,
                                               const float4 nh) {
  const float dx = rr.cx - cR.x, dy = rr.cy - cR.y, dz = rr.cz - cR.z;
  const float r2 = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
  const float s0 = __fmaf_rn(nh.z, dz, __fmaf_rn(nh.y, dy, __fmul_rn(nh.x, dx)));
  const float e = 1e-6f * ((fabsf(dx) + fabsf(dy)) + fabsf(dz));  // rounding of the dot product
  const float as0 = fabsf(s0);
  const float a_hi = as0 + e;
  const float b2 = fmaxf(__fmaf_rn(r2, 0.999999f, -(a_hi * a_hi) * 1.000003f), 0.0f);
  const float gt = fmaxf(__fmaf_rn(so_sqrt(b2), 0.999998f, -(rr.rS + cR.w)), 0.0f);
  const float alpha = fminf(fmaxf(__fmaf_rn(nh.z, ngz, __fmaf_rn(nh.y, ngy, __fmul_rn(nh.x, ngx))), -1.0f), 1.0f);
  const float mx = __fmaf_rn(-alpha, ngx, nh.x), my = __fmaf_rn(-alpha, ngy, nh.y), mz = __fmaf_rn(-alpha, ngz, nh.z);
  const float mlen = so_sqrt(__fmaf_rn(mz, mz, __fmaf_rn(my, my, __fmul_rn(mx, mx)))) * 1.000002f + 2e-6f;
  const float beta = s0 < 0.0f ? -alpha : alpha;
  const float ab = fabsf(beta);
  const float U = beta < 0.0f ? rr.Um : rr.Up;
  float reach = __fmaf_rn(ab, U, (1.0f - ab) * rr.rho) - (as0 - e) + __fmaf_rn(mlen, rr.rS, nh.w);
  reach += 4e-6f * ((((rr.rho + as0) + (rr.rS + nh.w)) + fabsf(rr.Up)) + fabsf(rr.Um));
  reach = fminf(reach, 1e30f);  // degenerate discs carry FLT_MAX: keep the product below finite
  return reach >= 0.0f && gt * gt <= 2.00002f * rr.rho * reach;
}

// point_disc_lb (traverse.hpp) for the TWO discs of a union pair block at once: the same operations component by
// component on packed registers (v_pk_add / v_pk_mul / v_pk_fma_f32), with so_sqrt, and the rounding allowance of the
// dot product taken from the largest |component| (3e-6 max >= 1e-6 sum).  A pair block is 64 bytes, the components of
// slots 2p and 2p + 1 interleaved so that every packed operand is one register pair of a 16-byte read:
//   P0 = (cx0 cx1 cy0 cy1)  P1 = (cz0 cz1 R0 R1)  P2 = (nx0 nx1 ny0 ny1)  P3 = (nz0 nz1 hn0 hn1)
__device__ __forceinline__ v2f so_disc_lb2(float qx, float qy, float qz, const float4 P0, const float4 P1, const float4 P2,
                                           const float4 P3) {
  const v2f dx = v2f{qx, qx} - v2f{P0.x, P0.y}, dy = v2f{qy, qy} - v2f{P0.z, P0.w}, dz = v2f{qz, qz} - v2f{P1.x, P1.y};
  const v2f r2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
  const v2f dot = __builtin_elementwise_fma(v2f{P3.x, P3.y}, dz, __builtin_elementwise_fma(v2f{P2.z, P2.w}, dy, v2f{P2.x, P2.y} * dx));
  const v2f a = {fabsf(dot.x), fabsf(dot.y)};
  const v2f e = v2f{fmaxf(fmaxf(fabsf(dx.x), fabsf(dy.x)), fabsf(dz.x)), fmaxf(fmaxf(fabsf(dx.y), fabsf(dy.y)), fabsf(dz.y))} * 3e-6f;
  const v2f a_hi = a + e;  // >= |n.q'|
  const v2f t = (a_hi * a_hi) * -1.000003f;
  const v2f b2 = __builtin_elementwise_fma(r2, v2f{0.999999f, 0.999999f}, t);
  const v2f sq = {so_sqrt(fmaxf(b2.x, 0.0f)), so_sqrt(fmaxf(b2.y, 0.0f))};
  const v2f gtr = __builtin_elementwise_fma(sq, v2f{0.999998f, 0.999998f}, -v2f{P1.z, P1.w});
  const v2f gnr = (a - e) - v2f{P3.z, P3.w};
  const v2f gt = {fmaxf(gtr.x, 0.0f), fmaxf(gtr.y, 0.0f)}, gn = {fmaxf(gnr.x, 0.0f), fmaxf(gnr.y, 0.0f)};
  return __builtin_elementwise_fma(gt, gt, gn * gn) * DISC_SHRINK;
}

// LDS-DMA of up to 16 leaf blocks (x[16] y[16] z[16]: twelve 16-byte chunks each) into a 3 KB staging area,
// transposed: chunk c of the leaf in slot s lands at ((c * 16 + s) * 16) bytes (see traverse(): SPARSE).
// `leaf_id`: the leaf of slot (lane & 15), NO_INDEX for an empty slot.
__device__ __forceinline__ void so_stage(const IndexView& ix, float* dst, uint32_t leaf_id) {
  const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (leaf_id != NO_INDEX) {
      const float* src = ix.soa + size_t(leaf_id) * LEAF_FLOATS + (i * 4 + (lane >> 4)) * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + i * (WAVE * 4)), 16, 0, 0);
    }
  }
}

// LDS-DMA of up to 8 leaf blocks, slot-minor with stride 8: chunk c of the leaf in slot s lands at ((c * 8 + s) * 16)
// bytes of a 1.5 KB area (read with NN1MinT::leaf_at<8>).  `leaf_id`: the leaf of slot (lane & 7).
__device__ __forceinline__ void so_stage8(const IndexView& ix, float* dst, uint32_t leaf_id) {
  const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i * 8 + (lane >> 3);
    if (leaf_id != NO_INDEX && c < 12) {
      const float* src = ix.soa + size_t(leaf_id) * LEAF_FLOATS + c * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + i * (WAVE * 4)), 16, 0, 0);
    }
  }
}

// One target point near (Cx, Cy, Cz), found by the whole wave: DOWN the hierarchy once, at every level into the child whose
// box is nearest (no ranking, no stack: 4 scans at 10M points), then the nearest of that leaf's 16 points.  Any point of the
// index is a valid seed -- it only gives the lanes a radius, the search that follows is exact whatever it is.  Used by the
// stand-off search for groups without a predecessor, and by the seeded search when every seed of a group is far away (the
// second iteration of an alignment slides the queries tens of point spacings along the surface).  All 64 lanes call it.
__device__ __forceinline__ uint32_t wave_greedy_point(const IndexView& ix, float Cx, float Cy, float Cz, const Box* topbox,
                                                      TraverseStats& ts) {
  const int lane = threadIdx.x & (WAVE - 1);
  const float INF = __builtin_inff();
  uint32_t seed_pos = NO_INDEX;
  uint32_t level = uint32_t(ix.top) + 1u, node = 0u;
  while (level > 1u) {
    const uint32_t cl = level - 1u, first = node * FANOUT;
    const Box* level_box = ix.box[1];
    uint32_t total = ix.count[1], coff = 0;
#pragma unroll
    for (int l = 2; l < MAX_LEVELS; ++l) {
      if (cl == uint32_t(l)) {
        level_box = ix.box[l];
        total = ix.count[l];
        coff = ix.cache_off[l];
      }
    }
    const uint32_t nchild = (total - first) < uint32_t(FANOUT) ? (total - first) : uint32_t(FANOUT);
    const bool has = uint32_t(lane) < nchild;
    float lb = INF;
    if (has) {
      const Box b = (int(cl) >= ix.cache_from) ? topbox[coff + first + lane] : level_box[first + lane];
      lb = point_box_lb(Cx, Cy, Cz, b.lo.x, b.lo.y, b.lo.z, b.hi.x, b.hi.y, b.hi.z);
    }
    const float m = wave_min_f(lb);
    const uint64_t at = __builtin_amdgcn_ballot_w64(has && lb == m);
    node = first + uint32_t(at != 0 ? __builtin_ctzll(at) : 0);
    level = cl;
    SO_COUNT(++ts.c[0]);
  }
  float d = INF;
  if (lane < LEAF) {
    const float4 t = ix.pts[size_t(node) * LEAF + lane];
    d = l2_simple(Cx, Cy, Cz, t.x, t.y, t.z);  // pad slots hold FLT_MAX sentinels: +inf
  }
  const float m = wave_min_f(d);
  const uint64_t at = __builtin_amdgcn_ballot_w64(lane < LEAF && d == m && d < INF);
  if (at != 0) seed_pos = node * LEAF + uint32_t(__builtin_ctzll(at));
  return seed_pos;
}

// Where the seed comes from: the previous group of this wave (every lane's query as searched and its match), and a
// history of one (query, match) pair of each of the wave's last 64 groups, one per lane -- consecutive groups of the
// kd order are usually neighbours, but a quarter of the steps jump, and then an older group is the nearer one.
struct PrevGroup {
  float x, y, z;
  uint32_t pos;  // sorted position of the lane's match, NO_INDEX if none
  float hx, hy, hz;
  uint32_t hpos;  // history entry of this lane
  uint32_t count; // groups recorded so far (wave-uniform)
  // Back-off: where the stand-off is so wide (or the surface so thick against the spacing) that the lists outgrow the
  // LDS, every attempt costs a collect that is thrown away.  Three give-ups in a row and the next six groups of the wave
  // go straight to traverse() -- still seeded, which is most of what a tight start buys there -- before it tries again.
  uint32_t fail_streak, skip;
  bool attempted;  // the last standoff_search went past the back-off gate (only those count as give-ups)
  __device__ __forceinline__ void init() {
    x = y = z = hx = hy = hz = 0.0f;
    pos = hpos = NO_INDEX;
    count = 0;
    fail_streak = 0;
    skip = 0;
    attempted = false;
  }
  // the wave jumps to another part of the cloud: its earlier matches are no seeds there (the back-off state stays)
  __device__ __forceinline__ void forget() {
    pos = hpos = NO_INDEX;
    count = 0;
  }
  __device__ __forceinline__ bool skip_now() {  // wave-uniform
    attempted = skip == 0;
    if (skip == 0) return false;
    --skip;
    return true;
  }
  __device__ __forceinline__ void outcome(bool done) {
    if (!attempted) return;
    attempted = false;
    if (done) {
      fail_streak = 0;
    } else if (++fail_streak >= 3u) {
      fail_streak = 0;
      skip = 6;
    }
  }
  // after a group: qx/qy/qz as searched, `mpos` the lane's match position or NO_INDEX
  __device__ __forceinline__ void record(float qx, float qy, float qz, uint32_t mpos) {
    x = qx;
    y = qy;
    z = qz;
    pos = mpos;
    const uint64_t fm = __builtin_amdgcn_ballot_w64(mpos != NO_INDEX);
    if (fm != 0) {
      // a lane from the middle of the group's kd order when it has a match, else the first that has
      const int src = (fm >> 32) & 1ull ? 32 : __builtin_ctzll(fm);
      const float rx = readlane_f(qx, src), ry = readlane_f(qy, src), rz = readlane_f(qz, src);
      const uint32_t rp = uint32_t(__builtin_amdgcn_readlane(int(mpos), src));
      if (uint32_t(threadIdx.x & (WAVE - 1)) == (count & 63u)) {
        hx = rx;
        hy = ry;
        hz = rz;
        hpos = rp;
      }
      ++count;
    }
  }
};

// Returns true when the group is done (pol holds every valid lane's exact minimum, up to the tie / resolve steps the
// caller runs anyway); false when the caller has to run traverse() -- pol then holds valid bounds and seeds.
// `from2`: wave radii (squared) up to this go to traverse() (next to the surface a disc excludes little a box does
// not).  `seed_leaf_out`: leaf of the seed (a start hint for the fallback).  Must be called by all 64 lanes.
template <class WL>
__device__ __forceinline__ bool standoff_search(const IndexView& ix, float qx, float qy, float qz, bool valid, NN1Min& pol,
                                                WL& wl, const Box* topbox, TraverseStats& ts, PrevGroup& prev,
                                                bool allow_skip, float from2, uint32_t& seed_leaf_out) {
  // the wave's LDS block, stage by stage (bytes): [0, 1664) two frontier buffers while collecting; [0, 3968) the disc
  // entries (32 B each) from the group cull on; [3968, 4224) ids of the union slots; [4224, 4992) ids of the group
  // cull's survivors; buf [4992, 8064) the collected ids, then the staging area of the evaluations
  static_assert(WL::BUF_FLOATS >= 768, "staging / id area: 3 KB");
  static_assert(sizeof(wl.stack) == 1664 && sizeof(wl.list) == 3072 && sizeof(wl.rad) == 256, "LDS plan of standoff.hpp");
  static_assert(offsetof(WL, list) == 1664 && offsetof(WL, rad) == 4736, "LDS plan of standoff.hpp");
  const int lane = threadIdx.x & (WAVE - 1);
  const uint32_t sub = uint32_t(lane) & 15u;
  const float BIG = 3.402823466e+38f;
  const float INF = __builtin_inff();
  seed_leaf_out = NO_INDEX;
  if (__builtin_amdgcn_ballot_w64(valid) == 0 || ix.n == 0) return true;
  const float qxa[1] = {qx}, qya[1] = {qy}, qza[1] = {qz};
#ifdef PCLHIP_SO_PROFILE
  uint64_t so_t = clock64();
#endif
  // ---- group box -------------------------------------------------------------------------------------------------
  float lx0 = valid ? qx : BIG, ly0 = valid ? qy : BIG, lz0 = valid ? qz : BIG;
  float hx0 = valid ? qx : -BIG, hy0 = valid ? qy : -BIG, hz0 = valid ? qz : -BIG, dummy = 0.0f;
  wave_min3_max4(lx0, ly0, lz0, hx0, hy0, hz0, dummy);
  const float Qlx = lx0, Qly = ly0, Qlz = lz0, Qhx = hx0, Qhy = hy0, Qhz = hz0;
  const float Cx = 0.5f * (Qlx + Qhx), Cy = 0.5f * (Qly + Qhy), Cz = 0.5f * (Qlz + Qhz);

  // ---- 0. seed ---------------------------------------------------------------------------------------------------
  uint32_t seed_pos = NO_INDEX;
  {
    const bool pok = prev.pos != NO_INDEX, hok = prev.hpos != NO_INDEX;
    if (__builtin_amdgcn_ballot_w64(pok || hok) != 0) {
      const float dx = prev.x - Cx, dy = prev.y - Cy, dz = prev.z - Cz;
      const float ex = prev.hx - Cx, ey = prev.hy - Cy, ez = prev.hz - Cz;
      const float d1 = pok ? (dx * dx + dy * dy) + dz * dz : INF;
      const float d2 = hok ? (ex * ex + ey * ey) + ez * ez : INF;
      const float d = fminf(d1, d2);
      const uint32_t cand = d1 <= d2 ? prev.pos : prev.hpos;
      const float m = wave_min_f(d);
      const uint64_t at = __builtin_amdgcn_ballot_w64((pok || hok) && d == m);
      if (at != 0) seed_pos = uint32_t(__builtin_amdgcn_readlane(int(cand), __builtin_ctzll(at)));
    }
    if (seed_pos == NO_INDEX) {
      // No previous group (the first of a run) or none with a match.  Any target point near the group will do as a seed --
      // it only has to give every lane a radius, the search below is exact whatever it is -- so instead of one lane's exact
      // neighbour (an unseeded best-first search that the whole wave executes: ~2000 wave-instructions, a quarter of a
      // group's own search, for one group in four) the wave walks DOWN the hierarchy once, at every level into the child
      // whose box is nearest to the group's centre (4 scans without ranking at 10M points), and takes that leaf's point
      // nearest to the centre.  (Round 4, 10M points: cold launch 2.00 -> 1.80 ms.)
      seed_pos = wave_greedy_point(ix, Cx, Cy, Cz, topbox, ts);
    }
    if (seed_pos == NO_INDEX) {
      // no previous group (first of the wave's chunk) or none with a match: the exact neighbour of ONE lane
      const uint64_t vm = __builtin_amdgcn_ballot_w64(valid);
      const bool one[1] = {valid && lane == __builtin_ctzll(vm)};
#if defined(PCLHIP_SO_PROFILE) || defined(PCLHIP_SO_REASONS)
      TraverseStats seed_ts;
#else
      TraverseStats& seed_ts = ts;
#endif
      traverse<NN1Min, true>(ix, qxa, qya, qza, one, pol, wl, topbox, seed_ts, NO_INDEX, true);
      pol.resolve(ix, qxa, qya, qza);
      seed_pos = uint32_t(__builtin_amdgcn_readlane(int(pol.bestpos[0]), __builtin_ctzll(vm)));
    }
  }
  if (seed_pos != NO_INDEX) {
    seed_leaf_out = seed_pos / LEAF;
    const float4 t = ix.pts[seed_pos];  // wave-uniform address
    if (valid) pol.seed(0, l2_simple(qx, qy, qz, t.x, t.y, t.z), seed_pos);
  }
  const float T = wave_max_f(valid ? pol.worst(0) : 0.0f);  // wave radius (squared); fixed while collecting
  SO_LAP(0);
  SO_MARK("seed_end");
  if (!(T < INF)) {
    SO_WHY(0);
    return false;
  }
  if (!(T > from2)) {
    SO_WHY(1);
    return false;
  }
  if (prev.skip_now()) return false;  // backing off: the lanes are seeded, traverse() does the rest

  char* const lds = reinterpret_cast<char*>(&wl);
  uint32_t* const ids = reinterpret_cast<uint32_t*>(wl.buf);
  float4* const dl = reinterpret_cast<float4*>(lds);  // disc entry e: dl[2e] = (centre, R), dl[2e+1] = (normal, hn)
  uint32_t* const idu = reinterpret_cast<uint32_t*>(lds + 3968);
  uint32_t* const sid = reinterpret_cast<uint32_t*>(lds + 4224);
  static_assert(32 * SO_DISC_CAP == 3968 && 3968 + 4 * SO_UNION_CAP == 4224 && 4224 + 4 * SO_SURV_CAP == 4992, "LDS plan");
  static_assert(offsetof(WL, buf) == 4992, "LDS plan");

  // ---- 1. collect: breadth first, the box rows of up to four nodes in flight together ------------------------------
  uint32_t n = 0;
  {
    uint32_t* fr_cur = reinterpret_cast<uint32_t*>(lds);
    uint32_t* fr_next = reinterpret_cast<uint32_t*>(lds + 832);
    constexpr uint32_t FR_CAP = 208;
    uint32_t level = uint32_t(ix.top) + 1u, node0 = 0u;  // virtual root above the top level
    if (allow_skip && seed_leaf_out != NO_INDEX) {        // start below the root when the search ball fits (see traverse())
      const auto inside = [&](const Box& b) {
        const float d = fminf(fminf(fminf(Qlx - b.lo.x, b.hi.x - Qhx), fminf(Qly - b.lo.y, b.hi.y - Qhy)),
                              fminf(Qlz - b.lo.z, b.hi.z - Qhz));
        return d > 0.0f && d * d * 0.999999f > T;
      };
      const bool has2 = ix.top >= 2, has3 = ix.top >= 3;
      Box b2 = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)}, b3 = b2;
      if (has2) b2 = ix.cell2 != nullptr ? ix.cell2[seed_leaf_out >> 6] : ix.box[2][seed_leaf_out >> 6];
      if (has3) b3 = ix.cell3 != nullptr ? ix.cell3[seed_leaf_out >> 12] : ix.box[3][seed_leaf_out >> 12];
      if (has2 && inside(b2)) {
        level = 2u;
        node0 = seed_leaf_out >> 6;
      } else if (has3 && inside(b3)) {
        level = 3u;
        node0 = seed_leaf_out >> 12;
      }
    }
    if (lane == 0) fr_cur[0] = node0;
    uint32_t ncur = 1;
    for (;;) {
      __builtin_amdgcn_wave_barrier();
      ncur = uniform_u32(ncur);
      const uint32_t cl = level - 1u;  // level of the children
      const Box* level_box = ix.box[1];
      uint32_t total = ix.count[1], coff = 0;
#pragma unroll
      for (int l = 2; l < MAX_LEVELS; ++l) {
        if (cl == uint32_t(l)) {
          level_box = ix.box[l];
          total = ix.count[l];
          coff = ix.cache_off[l];
        }
      }
      const bool cached = int(cl) >= ix.cache_from;  // upper levels: boxes come from the block's LDS copy
      uint32_t nnext = 0;
      for (uint32_t k = 0; k < ncur; k += 2u) {
        // two nodes per step: both box rows are in flight together (unconditional loads, clamped indices)
        const bool two = k + 1u < ncur;
        uint32_t first[2], nchild[2];
        Box b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          first[j] = uniform_u32(fr_cur[(j == 1 && two) ? k + 1u : k]) * FANOUT;
          nchild[j] = (total - first[j]) < uint32_t(FANOUT) ? (total - first[j]) : uint32_t(FANOUT);
          const uint32_t at = first[j] + (uint32_t(lane) < nchild[j] ? uint32_t(lane) : nchild[j] - 1u);
          b[j] = cached ? topbox[coff + at] : level_box[at];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (j == 0 || two) {
            SO_COUNT(++ts.c[0]);
            const bool has = uint32_t(lane) < nchild[j];
            const float lbG = box_box_lb(Qlx, Qly, Qlz, Qhx, Qhy, Qhz, b[j].lo.x, b[j].lo.y, b[j].lo.z, b[j].hi.x, b[j].hi.y,
                                         b[j].hi.z);
            const bool alive = has && !(lbG > T);
            const uint64_t mask = __builtin_amdgcn_ballot_w64(alive);
            if (mask != 0) {
              const uint32_t cnt = uint32_t(__builtin_popcountll(mask));
              const uint32_t pre = uint32_t(__builtin_popcountll(mask & ((1ull << lane) - 1ull)));
              if (cl == 1u) {
                if (n + cnt > SO_LIST_CAP) {
                  SO_WHY(2);
                  return false;
                }
                if (alive) ids[n + pre] = first[j] + uint32_t(lane);
                n += cnt;
              } else {
                if (nnext + cnt > FR_CAP) {
                  SO_WHY(3);
                  return false;
                }
                if (alive) fr_next[nnext + pre] = first[j] + uint32_t(lane);
                nnext += cnt;
                SO_COUNT(ts.c[3] += cnt);
              }
            }
          }
        }
      }
      if (cl == 1u || nnext == 0) break;
      uint32_t* const t2 = fr_cur;
      fr_cur = fr_next;
      fr_next = t2;
      ncur = nnext;
      level = cl;
    }
  }
  __builtin_amdgcn_wave_barrier();
  n = uniform_u32(n);
  SO_LAP(1);
  SO_MARK("collect_end");
  if (n == 0) {  // nothing within the wave radius (a finite maximum distance and no seed inside it)
    ++ts.c[4];
    return true;
  }

  // ---- 2. group cull: lane j <-> collected leaf l0 + j; the leaves that pass the reach filter of the WHOLE group (the
  // same bound with the maxima over all 64 lanes) are kept: ids in `sid`, the discs of the first SO_DISC_CAP of them in
  // LDS (`dl`) -- every later stage reads them from there
  const RowGeom geo = row_geometry(qx, qy, qz, valid);
  const uint32_t row = uint32_t(lane) >> 4;
  uint32_t ms = 0;  // survivors
  {
    RowReach gq;
    gq.cx = Cx;
    gq.cy = Cy;
    gq.cz = Cz;
    const float ex = Qhx - Qlx, ey = Qhy - Qly, ez = Qhz - Qlz;
    gq.rS = so_sqrt((ex * ex + ey * ey) + ez * ez) * 0.5000005f + 1e-6f * ((fabsf(Cx) + fabsf(Cy)) + fabsf(Cz));
    const float a_own = (geo.ngx * (qx - Cx) + geo.ngy * (qy - Cy)) + geo.ngz * (qz - Cz);
    const float rho = so_sqrt(valid ? pol.worst(0) : 0.0f) * 1.000004f;
    gq.Up = wave_max_f(valid ? rho - a_own : -BIG);
    gq.Um = wave_max_f(valid ? rho + a_own : -BIG);
    gq.rho = wave_max_f(valid ? rho : 0.0f);
    for (uint32_t l0 = 0; l0 < n; l0 += WAVE) {
      const uint32_t e = l0 + uint32_t(lane);
      const uint32_t id = ids[e < n ? e : n - 1u];
      const float4 cR = ix.disc[2 * size_t(id)], nh = ix.disc[2 * size_t(id) + 1];
      const bool al = e < n && (!(gq.rho < 1e30f) || so_reach_alive(gq, geo.ngx, geo.ngy, geo.ngz, cR, nh));
      const uint64_t bal = __builtin_amdgcn_ballot_w64(al);
#ifdef PCLHIP_VERIFY_BOUNDS  // collected leaves the group-level reach filter dropped: every lane checks every one of them
      for (uint64_t db = __builtin_amdgcn_ballot_w64(e < n && !al); db != 0; db &= db - 1ull) {
        const uint32_t vid = uint32_t(__builtin_amdgcn_readlane(int(id), __builtin_ctzll(db)));
        PCLHIP_VERIFY_CULL(ix.soa, vid, qx, qy, qz, pol.worst(0), valid, ts);
      }
#endif
      if (bal == 0) continue;
      const uint32_t cnt = uint32_t(__builtin_popcountll(bal));
      if (ms + cnt > SO_SURV_CAP) {
        SO_WHY(5);
        return false;
      }
      const uint32_t at = ms + uint32_t(__builtin_popcountll(bal & ((1ull << lane) - 1ull)));
      if (al) {
        sid[at] = id;
        if (at < SO_DISC_CAP) {
          dl[2 * at] = cR;
          dl[2 * at + 1] = nh;
        }
      }
      ms += cnt;
    }
  }
  __builtin_amdgcn_wave_barrier();
  ms = uniform_u32(ms);
  SO_LAP(2);
  SO_MARK("gcull_end");

  uint32_t id1 = NO_INDEX, id2 = NO_INDEX;  // the row's seed leaves (evaluated in the first batch)
  uint32_t stat_union = 0;
  // the survivors in batches of <= SO_DISC_CAP discs (one batch but for the widest stand-offs), every batch in segments
  // of <= 64 union slots
  for (uint32_t b0 = 0; b0 < ms; b0 += SO_DISC_CAP) {
    const uint32_t m = (ms - b0) < SO_DISC_CAP ? (ms - b0) : SO_DISC_CAP;
    if (b0 != 0) {  // later batches: their discs were not kept
      for (uint32_t e0 = 0; e0 < m; e0 += WAVE) {
        const uint32_t e = e0 + uint32_t(lane);
        if (e < m) {
          const uint32_t id = sid[b0 + e];
          dl[2 * e] = ix.disc[2 * size_t(id)];
          dl[2 * e + 1] = ix.disc[2 * size_t(id) + 1];
        }
      }
      __builtin_amdgcn_wave_barrier();
    } else {
      // ---- 3. row seeds (first batch): every row evaluates the two surviving leaves nearest to its centre ----------
      uint32_t m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu;  // the lane's two smallest (distance | entry) keys
      for (uint32_t e0 = 0; e0 < m; e0 += 16u) {
        const uint32_t e = e0 + sub;
        if (e < m) {
          const float4 c = dl[2 * e];
          const float dx = geo.cx - c.x, dy = geo.cy - c.y, dz = geo.cz - c.z;
          const float d = (dx * dx + dy * dy) + dz * dz;  // >= 0: its bit pattern orders like the value
          const uint32_t key = (__float_as_uint(d) & ~0x7Fu) | e;  // e < SO_DISC_CAP <= 128
          m2 = min(m2, max(m1, key));
          m1 = min(m1, key);
        }
      }
      const uint32_t r1 = row_min_u32(m1);
      const uint32_t r2 = row_min_u32(m1 == r1 ? m2 : m1);
      if (geo.any && r1 != 0xFFFFFFFFu) id1 = sid[r1 & 0x7Fu];
      if (geo.any && r2 != 0xFFFFFFFFu) id2 = sid[r2 & 0x7Fu];
      // slot 2 * row + k <- the row's k-th seed leaf; the slot ids travel through the idle union-id area
      if (sub == 0u) {
        idu[2 * row] = id1;
        idu[2 * row + 1] = id2;
      }
      __builtin_amdgcn_wave_barrier();
      so_stage8(ix, wl.buf, idu[lane & 7]);
      PCLHIP_WAIT_VMCNT0();
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (uint32_t k = 0; k < 2; ++k) {
        const uint32_t id = valid ? (k == 0 ? id1 : id2) : NO_INDEX;
        if (__builtin_amdgcn_ballot_w64(id != NO_INDEX) != 0) {
          SO_COUNT(++ts.c[2]);
          pol.template leaf_at<8>(reinterpret_cast<const float4*>(wl.buf) + (2 * row + k), id, qxa, qya, qza);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    SO_LAP(3);
    SO_MARK("rowseed_end");

    for (uint32_t c = 0; c < m;) {
      // ---- 4. row cull: (row, leaf) pairs, 64 per pass; leaves alive for any row -> union slots, compacted in place
      const RowReach rr = row_reach_of(geo, qx, qy, qz, valid, pol.worst(0));  // the radii of NOW
      uint32_t ucnt = 0;
      uint64_t rowmask;
      {
        uint32_t acc_lo = 0, acc_hi = 0;  // this lane's own alive pairs as union-slot bits
        for (; c < m && ucnt + 16u <= SO_UNION_CAP; c += 16u) {
          const uint32_t e = c + sub, ec = e < m ? e : m - 1u;
          const float4 cR = dl[2 * ec], nh = dl[2 * ec + 1];
          const uint32_t id = sid[b0 + ec];
          const bool al = e < m && geo.any && (!(rr.rho < 1e30f) || so_reach_alive(rr, geo.ngx, geo.ngy, geo.ngz, cR, nh));
          const uint64_t bal = __builtin_amdgcn_ballot_w64(al);
#ifdef PCLHIP_VERIFY_BOUNDS  // (row, leaf) pairs the row-level reach filter dropped: the lanes of that row check the leaf
          for (uint64_t db = __builtin_amdgcn_ballot_w64(e < m && geo.any && !al); db != 0; db &= db - 1ull) {
            const uint32_t bit = uint32_t(__builtin_ctzll(db));
            const uint32_t vid = uint32_t(__builtin_amdgcn_readlane(int(id), int(bit)));
            PCLHIP_VERIFY_CULL(ix.soa, vid, qx, qy, qz, pol.worst(0), valid && (uint32_t(lane) >> 4) == (bit >> 4), ts);
          }
#endif
          __builtin_amdgcn_wave_barrier();  // the pass has read its 16 entries before slots <= them are rewritten
          if (bal == 0) continue;
          const uint32_t any16 = uint32_t((bal | (bal >> 16) | (bal >> 32) | (bal >> 48)) & 0xFFFFull);
          const uint32_t u = ucnt + uint32_t(__builtin_popcount(any16 & ((1u << sub) - 1u)));
          // the same entry is held by one lane of every row: the lowest alive row writes the slot (u <= e: in place)
          const uint64_t below = bal & ((1ull << lane) - 1ull) & (0x0001000100010001ull << sub);
          if (al && below == 0) {  // pair block u / 2, half u & 1 (see so_disc_lb2)
            float* const w = reinterpret_cast<float*>(lds) + 16u * (u >> 1) + (u & 1u);
            w[0] = cR.x;
            w[2] = cR.y;
            w[4] = cR.z;
            w[6] = cR.w;
            w[8] = nh.x;
            w[10] = nh.y;
            w[12] = nh.z;
            w[14] = nh.w;
            idu[u] = id;
          }
          if (al) {  // u < 64
            const uint64_t bit = 1ull << u;
            acc_lo |= uint32_t(bit);
            acc_hi |= uint32_t(bit >> 32);
          }
          ucnt += uint32_t(__builtin_popcount(any16));
        }
        rowmask = (uint64_t(row_or_u32(acc_hi)) << 32) | uint64_t(row_or_u32(acc_lo));
      }
      __builtin_amdgcn_wave_barrier();
      ucnt = uniform_u32(ucnt);  // wave-uniform by construction: keep it in a scalar register
      SO_LAP(5);
      SO_MARK("rowcull_end");
      stat_union += ucnt;
      if (ucnt == 0) continue;

#if PCLHIP_SO_BATCHCULL
      // ---- 5 + 6. lane cull and evaluation, 16 union slots at a time: every lane runs the disc bound against the slots
      // alive for its row -- a pair block per step (packed math), against its radius of NOW, which the batches before have
      // tightened -- and evaluates what it could not exclude.  The batch's leaves are on their way into the staging area
      // while the tests run.  (The walk over the pair blocks is wave-uniform: a row keeps most of the union alive at a
      // stand-off, so skipping per lane would cost more mask arithmetic than the tests it saves.)
      for (uint32_t c0 = 0; c0 < ucnt; c0 += LEAF_BATCH) {
        const uint32_t rm16 = valid ? uint32_t((rowmask >> c0) & 0xFFFFull) : 0u;
        if (__builtin_amdgcn_ballot_w64(rm16 != 0) == 0) continue;
        const uint32_t cn = (ucnt - c0) < uint32_t(LEAF_BATCH) ? (ucnt - c0) : uint32_t(LEAF_BATCH);
        so_stage(ix, wl.buf, sub < cn ? idu[c0 + sub] : NO_INDEX);
        uint32_t m16 = 0;
        const float w = pol.worst(0);
        for (uint32_t b = 0; b < cn; b += 2u) {
          SO_COUNT(++ts.c[1]);
          const uint32_t rbits = (rm16 >> b) & 3u;
          if (__builtin_amdgcn_ballot_w64(rbits != 0) == 0) continue;
          const float4* const P = dl + 2u * (c0 + b);  // 4 float4 per pair block
          const v2f lb = so_disc_lb2(qx, qy, qz, P[0], P[1], P[2], P[3]);
          const uint2 ii = *reinterpret_cast<const uint2*>(idu + c0 + b);
          const bool need0 = (rbits & 1u) != 0 && !(lb.x > w) && ii.x != id1 && ii.x != id2;  // the row seeds are done
          const bool need1 = (rbits & 2u) != 0 && !(lb.y > w) && ii.y != id1 && ii.y != id2;
          PCLHIP_VERIFY_CULL(ix.soa, ii.x, qx, qy, qz, w, (rbits & 1u) != 0 && lb.x > w, ts);
          PCLHIP_VERIFY_CULL(ix.soa, ii.y, qx, qy, qz, w, (rbits & 2u) != 0 && lb.y > w, ts);
          m16 |= ((need0 ? 1u : 0u) | (need1 ? 2u : 0u)) << b;
        }
        SO_LAP(6);
        PCLHIP_WAIT_VMCNT0();
        while (__builtin_amdgcn_ballot_w64(m16 != 0) != 0) {
          uint32_t slot = 0, id = NO_INDEX;
          if (m16 != 0) {
            slot = uint32_t(__builtin_ctz(m16));
            id = idu[c0 + slot];
            m16 &= m16 - 1u;
          }
          SO_COUNT(++ts.c[2]);
          pol.leaf_lane(wl.buf, slot, id, qxa, qya, qza);
        }
        SO_LAP(7);
      }
#else
      // ---- 5. lane cull: every lane runs the disc bound against the union slots alive for its row, a pair block per
      // step (packed math); the walk is wave-uniform -- a row keeps most of the union alive at a stand-off, so skipping
      // per lane would cost more mask arithmetic than the tests it saves
      uint32_t lm_lo = 0, lm_hi = 0;  // lanemask: bit u = this lane has to evaluate union slot u
      {
        const float w = pol.worst(0);
        const uint32_t rm_lo = valid ? uint32_t(rowmask) : 0u, rm_hi = valid ? uint32_t(rowmask >> 32) : 0u;
        for (uint32_t b = 0; b < ucnt; b += 2u) {  // b = 2p
          SO_COUNT(++ts.c[1]);
          const uint32_t rbits = ((b < 32u ? rm_lo : rm_hi) >> (b & 31u)) & 3u;
          if (__builtin_amdgcn_ballot_w64(rbits != 0) == 0) continue;
          const float4* const P = dl + 2u * b;  // 4 float4 per pair block
          const v2f lb = so_disc_lb2(qx, qy, qz, P[0], P[1], P[2], P[3]);
          const uint2 ii = *reinterpret_cast<const uint2*>(idu + b);
          const bool need0 = (rbits & 1u) != 0 && !(lb.x > w) && ii.x != id1 && ii.x != id2;  // the row seeds are done
          const bool need1 = (rbits & 2u) != 0 && !(lb.y > w) && ii.y != id1 && ii.y != id2;
          PCLHIP_VERIFY_CULL(ix.soa, ii.x, qx, qy, qz, w, (rbits & 1u) != 0 && lb.x > w, ts);
          PCLHIP_VERIFY_CULL(ix.soa, ii.y, qx, qy, qz, w, (rbits & 2u) != 0 && lb.y > w, ts);
          const uint32_t nb = ((need0 ? 1u : 0u) | (need1 ? 2u : 0u)) << (b & 31u);
          if (b < 32u) lm_lo |= nb;
          else lm_hi |= nb;
        }
      }
      const uint64_t lanemask = (uint64_t(lm_hi) << 32) | uint64_t(lm_lo);
      SO_LAP(6);
      SO_MARK("lanecull_end");

      // ---- 6. evaluation, 16 staged leaves at a time, every lane its own ---------------------------------------------
      for (uint32_t c0 = 0; c0 < ucnt; c0 += LEAF_BATCH) {
        uint32_t m16 = uint32_t((lanemask >> c0) & 0xFFFFull);
        if (__builtin_amdgcn_ballot_w64(m16 != 0) == 0) continue;
        const uint32_t cn = (ucnt - c0) < uint32_t(LEAF_BATCH) ? (ucnt - c0) : uint32_t(LEAF_BATCH);
        so_stage(ix, wl.buf, sub < cn ? idu[c0 + sub] : NO_INDEX);
        PCLHIP_WAIT_VMCNT0();
        while (__builtin_amdgcn_ballot_w64(m16 != 0) != 0) {
          uint32_t slot = 0, id = NO_INDEX;
          if (m16 != 0) {
            slot = uint32_t(__builtin_ctz(m16));
            id = idu[c0 + slot];
            m16 &= m16 - 1u;
          }
          SO_COUNT(++ts.c[2]);
          pol.leaf_lane(wl.buf, slot, id, qxa, qya, qza);
        }
      }
#endif
      SO_LAP(7);
      SO_MARK("eval_end");
    }
  }
  SO_WHY(7);
  ++ts.c[4];
#if !defined(PCLHIP_SO_PROFILE) && !defined(PCLHIP_SO_REASONS) && !defined(PCLHIP_VERIFY_BOUNDS)
  ++ts.c[5];
  ts.c[6] += n;
  ts.c[7] += stat_union;
#else
  (void)stat_union;
#endif
  return true;
}

}  // namespace pclhip
