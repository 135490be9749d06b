// traverse.hpp -- wavefront-cooperative exact nearest-neighbour traversal of the implicit wide BVH.
//
// One wavefront (64 lanes) owns 64 spatially compact queries (consecutive in kd order), one per
// lane.  Tree nodes are visited by the WAVE, not by lanes:
//   * interior node: lane j loads child box j (one coalesced 2 KB read), tests it against the
//     bounding box of the wave's 64 queries and the wave's pruning radius T = max_i worst_i
//     (DPP butterflies); survivors are pushed on a wave-uniform stack in LDS, nearest child first.
//   * leaf-level node: the surviving leaves are ranked (by distance to the query-group box while
//     bounds are still loose), their ids + boxes go into a small list in LDS, and their candidate
//     blocks (x[16] y[16] z[16] [w[16]]) are fetched for up to 16 leaves at a time with
//     `global_load_lds_dwordx4` -- the vector-memory path with its deep queues, straight into LDS, no
//     VGPRs -- WHILE the per-lane box tests run.  (The first version read each leaf through the scalar
//     cache: one dependent scalar-load miss per visited leaf was the whole kernel's critical path.)
//   * evaluation, two forms.  SPARSE (1-NN, top-K in registers): the batch is staged transposed and
//     every lane evaluates only the leaves its own bound cannot exclude, popping them in rounds
//     (`ds_read_b128` of its own slot + packed `v_pk_*` math).  Wave-uniform (heap / radius policies): a
//     leaf that some lane still needs is evaluated by every lane from broadcast reads.
//   * a seeded search may start below the root (see `start_leaf`).
// All bounds are exact in float: rounding is monotone and the bound uses the same operation order
// as the distance, so box_lb(q, box) <= l2_simple(q, c) for every c in the box, bit for bit.
// Distances follow FLANN's L2_Simple order ((dx*dx)+dy*dy)+dz*dz with no FMA contraction
// (call sites kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp:154-203).
#pragma once

#include "pclhip_internal.hpp"

namespace pclhip {

// worst case for n < 2^31 points (int32 indices): 134M leaves -> levels of 2.1M, 32768, 512 and 8 boxes;
// the virtual root's scan pushes <= 7, every interior node below it <= 63 siblings: 7 + 3 * 63 = 196
constexpr int STACK_ENTRIES = 208;
constexpr int LEAF_BATCH = 16;
#ifndef PCLHIP_PAIR_MODE
#define PCLHIP_PAIR_MODE 1
#endif
constexpr bool pair_mode = PCLHIP_PAIR_MODE != 0;  // A/B: -DPCLHIP_PAIR_MODE=0 walks whole disc lists the sequential way         // leaves staged in LDS at a time (16 x 256 B = 4 KB)
constexpr int LEAF_FLOATS = 4 * LEAF;  // x[16] y[16] z[16] w[16]

// Per-wavefront LDS working set: traversal stack (1.6 KB), ranked leaf list (3.25 KB), staged candidate
// blocks.  WaveLdsT<3072> (no room for the w[16] chunks) is for kernels whose policies never stage the
// original indices: 6.6 KB per wave, which lets a 4-wave block fit five times into a CU's 160 KB.
template <int BUF_BYTES>
struct __attribute__((aligned(16))) WaveLdsT {
  static constexpr int BUF_FLOATS = BUF_BYTES / 4;
  static constexpr int LIST_STRIDE = 3;  // float4 per ranked leaf
  uint2 stack[STACK_ENTRIES];
  float4 list[3 * FANOUT];  // per ranked leaf: seeded searches (lo.xyz, id) (hi.xyz, lbG) (-);
                            // loose searches (disc centre.xyz, id) (-, -, -, lbG) (disc normal.xyz, hn), R in rad[]
  float rad[FANOUT];
  float buf[BUF_FLOATS];
  __device__ __forceinline__ float* radii() { return rad; }
};
// The same for searches that never bound leaves by their discs (self-queries: the normals): two float4 per ranked leaf
// and no radii -- 1.25 KB less per wave.  traverse() reads LIST_STRIDE and never takes the disc branches with it.
template <int BUF_BYTES>
struct __attribute__((aligned(16))) WaveLdsBoxT {
  static constexpr int BUF_FLOATS = BUF_BYTES / 4;
  static constexpr int LIST_STRIDE = 2;
  uint2 stack[STACK_ENTRIES];
  float4 list[2 * FANOUT];
  float buf[BUF_FLOATS];
  __device__ __forceinline__ float* radii() { return buf; }  // never dereferenced (no discs with this layout)
};
typedef WaveLdsT<LEAF_BATCH * LEAF_FLOATS * 4> WaveLds;  // 4 KB of staging: x y z w chunks of 16 leaves

__device__ __forceinline__ float l2_simple(float qx, float qy, float qz, float cx, float cy, float cz) {
  const float dx = __fsub_rn(qx, cx), dy = __fsub_rn(qy, cy), dz = __fsub_rn(qz, cz);
  float r = __fmul_rn(dx, dx);
  r = __fadd_rn(r, __fmul_rn(dy, dy));
  r = __fadd_rn(r, __fmul_rn(dz, dz));
  return r;
}

// squared distance lower bound point <-> box, same op order as l2_simple
__device__ __forceinline__ float point_box_lb(float qx, float qy, float qz, float lx, float ly, float lz,
                                              float hx, float hy, float hz) {
  const float gx = fmaxf(fmaxf(__fsub_rn(lx, qx), __fsub_rn(qx, hx)), 0.0f);
  const float gy = fmaxf(fmaxf(__fsub_rn(ly, qy), __fsub_rn(qy, hy)), 0.0f);
  const float gz = fmaxf(fmaxf(__fsub_rn(lz, qz), __fsub_rn(qz, hz)), 0.0f);
  float r = __fmul_rn(gx, gx);
  r = __fadd_rn(r, __fmul_rn(gy, gy));
  r = __fadd_rn(r, __fmul_rn(gz, gz));
  return r;
}

// lower bound between the query-group box [Ql,Qh] and a node box [l,h]
__device__ __forceinline__ float box_box_lb(float Qlx, float Qly, float Qlz, float Qhx, float Qhy, float Qhz,
                                            float lx, float ly, float lz, float hx, float hy, float hz) {
  const float gx = fmaxf(fmaxf(__fsub_rn(lx, Qhx), __fsub_rn(Qlx, hx)), 0.0f);
  const float gy = fmaxf(fmaxf(__fsub_rn(ly, Qhy), __fsub_rn(Qly, hy)), 0.0f);
  const float gz = fmaxf(fmaxf(__fsub_rn(lz, Qhz), __fsub_rn(Qlz, hz)), 0.0f);
  float r = __fmul_rn(gx, gx);
  r = __fadd_rn(r, __fmul_rn(gy, gy));
  r = __fadd_rn(r, __fmul_rn(gz, gz));
  return r;
}

// ---- disc bounds -------------------------------------------------------------------------------------
// Every leaf also carries a bounded cylinder ("disc") that holds its points: centre c, radius R >= |p - c|,
// direction n with |n| <= 1 (least variance of the leaf's points) and half thickness hn >= |n.(p - c)|
// (index_build.hip: leaf_disc_kernel).  Splitting q - p into its parts along and across n,
//     |q - p|^2 >= max(|n.q'| - hn, 0)^2 + max(|q'_perp| - R, 0)^2,     q' = q - c,
// which couples the stand-off of a query from a sloped sheet with its offset along the sheet -- the axis-aligned
// box of a 16-point patch is ~10x thicker than the patch and cannot: a query 0.015 off the surface has to
// evaluate every leaf whose box it sees within sqrt(2 * 0.015 * thickness).  Unlike the AABB bound this one
// is not bit-monotone; it is made safe instead: every rounding below is covered by an explicit allowance
// (1e-6 relative, ~16 ulp, against 1-4 ulp of actual error), and the result is shrunk once more before it is
// compared with a float l2_simple distance.  Used for loose searches only; seeded ones keep the exact boxes.
constexpr float DISC_SHRINK = 0.999996f;
__device__ __forceinline__ float point_disc_lb(float qx, float qy, float qz, const float4 cR, const float4 nh) {
  const float dx = qx - cR.x, dy = qy - cR.y, dz = qz - cR.z;
  const float r2 = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
  const float a = fabsf(__fmaf_rn(nh.z, dz, __fmaf_rn(nh.y, dy, __fmul_rn(nh.x, dx))));
  const float e = 1e-6f * ((fabsf(dx) + fabsf(dy)) + fabsf(dz));  // rounding of the dot product
  const float a_hi = a + e;                                        // >= |n.q'|
  // across n: |q'|^2 - (n.q' / |n|)^2 with |n|^2 >= 1 - 1e-6 (the build shrinks n by 4e-7)
  const float b2 = fmaxf(__fmaf_rn(r2, 0.999999f, -(a_hi * a_hi) * 1.000003f), 0.0f);
  const float gt = fmaxf(__fmaf_rn(__fsqrt_rn(b2), 0.999999f, -cR.w), 0.0f);
  const float gn = fmaxf((a - e) - nh.w, 0.0f);
#ifdef PCLHIP_VERIFY_MUTATE  // a deliberately WRONG bound: proves that the PCLHIP_VERIFY_BOUNDS build sees a broken claim
  return __fmaf_rn(gt, gt, gn * gn) * PCLHIP_VERIFY_MUTATE;
#else
  return __fmaf_rn(gt, gt, gn * gn) * DISC_SHRINK;
#endif
}
// ---- row filter for disc leaves (pair mode of the loose path, see traverse()) -------------------------------
// A lower bound for a whole GROUP of queries standing h off a sheet takes the smallest stand-off against the largest
// radius, and every slack in it is multiplied by the stand-off: r^2 < 2 h slack keeps ~70 leaves alive per 64-query
// group where the lanes need ~16.  What a lane needs is decided by d_i - lb_i, in which h_i^2 cancels; the filter
// below therefore bounds the per-lane REACH over a set S of lanes (here: a row of 16).  With rho_i = sqrt(worst_i), a
// direction ng (any vector; the patch normal is the useful one), the set's centre C and radius rS, w_i = q_i - C,
// a_i = ng.w_i and the decomposition n = alpha ng + m of the disc normal:
//     sigma_i = n.(q_i - c) = s0 + alpha a_i + m.w_i,   s0 = n.(C - c),   |m.w_i| <= |m| rS
//     g_i = max(|sigma_i| - hn, 0)  is the lane's gap along n;  lane i needs the leaf only if  g_i <= rho_i  and
//     lat_i^2 <= rho_i^2 - g_i^2 = (rho_i - g_i)(rho_i + g_i) <= 2 rho_max (rho_i - g_i)
//     rho_i - g_i <= rho_i - sgn(s0) sigma_i + hn <= (rho_i - beta a_i) - |s0| + |m| rS + hn,   beta = sgn(s0) alpha
//     rho_i - beta a_i <= |beta| U(+/-) + (1 - |beta|) rho_max,   U+ = max_S (rho_i - a_i),  U- = max_S (rho_i + a_i)
// so a leaf whose lateral gap gt to the set (centre distance across n less both radii) has gt^2 > 2 rho_max reach, or
// whose reach is negative, is needed by no lane of S.  Every rounding is covered by explicit allowances as in
// point_disc_lb.  (scratch/standoff_model.py: 356 (row, leaf) pairs -> 64 alive per group at the bench's stand-off.)
struct RowReach {
  float cx, cy, cz, rS;  // centre and radius of the row's queries (radius rounded up)
  float Up, Um, rho;     // U+, U-, rho_max of the row (rounded up)
};
__device__ __forceinline__ bool row_reach_alive(const RowReach& rr, float ngx, float ngy, float ngz, const float4 cR,
                                                const float4 nh) {
  const float dx = rr.cx - cR.x, dy = rr.cy - cR.y, dz = rr.cz - cR.z;
  const float r2 = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
  const float s0 = __fmaf_rn(nh.z, dz, __fmaf_rn(nh.y, dy, __fmul_rn(nh.x, dx)));
  const float e = 1e-6f * ((fabsf(dx) + fabsf(dy)) + fabsf(dz));  // rounding of the dot product
  const float as0 = fabsf(s0);
  const float a_hi = as0 + e;
  // across n: |d|^2 - (n.d / |n|)^2 with |n|^2 >= 1 - 1e-6 (the build shrinks n by 4e-7)
  const float b2 = fmaxf(__fmaf_rn(r2, 0.999999f, -(a_hi * a_hi) * 1.000003f), 0.0f);
  const float gt = fmaxf(__fmaf_rn(__fsqrt_rn(b2), 0.999999f, -(rr.rS + cR.w)), 0.0f);
  const float alpha = fminf(fmaxf(__fmaf_rn(nh.z, ngz, __fmaf_rn(nh.y, ngy, __fmul_rn(nh.x, ngx))), -1.0f), 1.0f);
  const float mx = __fmaf_rn(-alpha, ngx, nh.x), my = __fmaf_rn(-alpha, ngy, nh.y), mz = __fmaf_rn(-alpha, ngz, nh.z);
  const float mlen = __fsqrt_rn(__fmaf_rn(mz, mz, __fmaf_rn(my, my, __fmul_rn(mx, mx)))) * 1.000001f + 2e-6f;
  const float beta = s0 < 0.0f ? -alpha : alpha;
  const float ab = fabsf(beta);
  const float U = beta < 0.0f ? rr.Um : rr.Up;
  float reach = __fmaf_rn(ab, U, (1.0f - ab) * rr.rho) - (as0 - e) + __fmaf_rn(mlen, rr.rS, nh.w);
  reach += 4e-6f * ((((rr.rho + as0) + (rr.rS + nh.w)) + fabsf(rr.Up)) + fabsf(rr.Um));
  reach = fminf(reach, 1e30f);  // degenerate discs carry FLT_MAX: keep the product below finite
  return reach >= 0.0f && gt * gt <= 2.00002f * rr.rho * reach;
}

// ---- wavefront reductions (all 64 lanes must be active) ---------------------------------------
// DPP butterflies (no LDS traffic, plain VALU latency): quad_perm swaps, row_half_mirror and
// row_mirror leave every row of 16 lanes holding its row result; row_bcast15/31 then fold the four
// rows into lane 63.  The result is read back with readlane so the compiler knows it is
// wave-uniform (SGPR): every branch on it is a scalar branch.
// The DPP modifier is folded into the min/max itself (one VALU instruction per butterfly step; the
// update_dpp builtin costs a copy, a v_mov_dpp, a canonicalising v_max and the min).  A DPP read of a
// VGPR written by the previous VALU instruction needs two wait states: s_nop 1 in the single-value
// chains, independent work in between in the 7-wide version.  Inputs are never NaN.
// The reductions themselves live in <pclhip_wave_reduce.hpp> (this directory: inline DPP assembly).  The CPU emulation of
// the test tier puts its own directory in front on the include path and supplies the same functions as shuffles
// (tests/wavesim/pclhip_wave_reduce.hpp) -- an include-path hook, no conditional compilation here.
}  // namespace pclhip
#include <pclhip_wave_reduce.hpp>
namespace pclhip {
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ float uniform_f32(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

constexpr uint64_t KEY_NONE = (uint64_t(0x7F800000u) << 32) | 0xFFFFFFFFull;  // (+inf, no index)
__device__ __forceinline__ uint64_t make_key(float d, uint32_t idx) {
  return (uint64_t(__float_as_uint(d)) << 32) | idx;
}
__device__ __forceinline__ float key_dist(uint64_t k) { return __uint_as_float(uint32_t(k >> 32)); }
__device__ __forceinline__ uint32_t key_index(uint64_t k) { return uint32_t(k); }

typedef float v2f __attribute__((ext_vector_type(2)));

// distances of candidate pair j of a staged leaf block (LDS, x[16] y[16] z[16] w[16])
__device__ __forceinline__ v2f pair_dist(const float* l, int j, v2f qx2, v2f qy2, v2f qz2) {
  const v2f* p = reinterpret_cast<const v2f*>(l);
  const v2f dx = qx2 - p[j], dy = qy2 - p[LEAF / 2 + j], dz = qz2 - p[LEAF + j];
  v2f r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}

// ---- leaf policies -------------------------------------------------------------------------------
// All policies see a staged leaf block `l` in LDS and the leaf id; every lane runs them.
//
// 1-NN, exact (distance, original index) order: used for ties and as the reference policy.
struct NN1 {
  uint64_t key;
  uint32_t pos;
  __device__ __forceinline__ void init(uint64_t k0) {
    key = k0;
    pos = NO_INDEX;
  }
  static constexpr int QPL = 1;
  __device__ __forceinline__ float worst(int) const { return key_dist(key); }
  __device__ __forceinline__ void leaf(const float* l, uint32_t leaf_id, const float* qx, const float* qy,
                                       const float* qz) {
    const uint32_t base = leaf_id * LEAF;
#pragma unroll
    for (int c = 0; c < LEAF; ++c) {
      const float d = l2_simple(qx[0], qy[0], qz[0], l[c], l[LEAF + c], l[2 * LEAF + c]);
      const uint64_t k = make_key(d, __float_as_uint(l[3 * LEAF + c]));
      const bool t = k < key;
      key = t ? k : key;
      pos = t ? base + c : pos;
    }
  }
  // lane-sparse evaluation from the transposed LDS staging buffer (see traverse(): SPARSE); this policy
  // only runs for the rare tie lanes, so the original indices come straight from the SoA copy (`soa`)
  // instead of occupying staging space
  static constexpr bool LANE_SPARSE = true;
  static constexpr bool NEEDS_W = false;
  const float* soa = nullptr;  // IndexView::soa
  __device__ __forceinline__ void leaf_lane(const float* buf, uint32_t slot, uint32_t leaf_id, const float* qx,
                                            const float* qy, const float* qz) {
    leaf_at<16>(reinterpret_cast<const float4*>(buf) + slot, leaf_id, qx, qy, qz);  // chunk c at s[c * 16]
  }
  // the same straight from the index's SoA copy (chunk c at s[c]): the per-lane search (lane_search.hpp)
  __device__ __forceinline__ void leaf_global(const float* soa_, uint32_t leaf_id, const float* qx, const float* qy,
                                              const float* qz) {
    leaf_at<1>(reinterpret_cast<const float4*>(soa_ + size_t(leaf_id != NO_INDEX ? leaf_id : 0u) * LEAF_FLOATS), leaf_id, qx, qy,
               qz);
  }
  template <int STRIDE>
  __device__ __forceinline__ void leaf_at(const float4* s, uint32_t leaf_id, const float* qx, const float* qy,
                                          const float* qz) {
    if (leaf_id != NO_INDEX) {
      const float4* wsrc = reinterpret_cast<const float4*>(soa + size_t(leaf_id) * LEAF_FLOATS + 3 * LEAF);
      const uint32_t base = leaf_id * LEAF;
#pragma unroll
      for (int c4 = 0; c4 < LEAF / 4; ++c4) {
        const float4 X = s[c4 * STRIDE], Y = s[(4 + c4) * STRIDE], Z = s[(8 + c4) * STRIDE], W = wsrc[c4];
        const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w}, zs[4] = {Z.x, Z.y, Z.z, Z.w},
                    ws[4] = {W.x, W.y, W.z, W.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = l2_simple(qx[0], qy[0], qz[0], xs[j], ys[j], zs[j]);
          const uint64_t k = make_key(d, __float_as_uint(ws[j]));
          const bool t = k < key;
          key = t ? k : key;
          pos = t ? base + uint32_t(4 * c4 + j) : pos;
        }
      }
    }
  }
};

// 1-NN fast path: the hot loop only tracks the minimum DISTANCE (packed v_pk_* math on candidate
// pairs, one v_min3 per pair) and the LEAF that first reached it (a wave-uniform id: one select).
// WHICH slot of that leaf won is resolved once per query after the traversal (resolve()): the lane
// re-reads its winning leaf and recomputes the same 16 distances.  Exact distance ties (two slots of
// a leaf, or an equal minimum in another leaf) raise `tie`; the caller then re-runs the exact policy
// NN1 for those lanes, so results stay bit-identical to the oracle.
template <int Q>
struct NN1MinT {
  static constexpr int QPL = Q;  // queries per lane: the wave owns 64*Q queries and every staged
                                 // candidate block / node scan / leaf test is shared by all of them
  float best[Q];         // candidates must be strictly below this to win
  uint32_t bestpos[Q];   // sorted position of the winner; while `unres`: first slot of the winning leaf;
                         // NO_INDEX while best is only the bound
  bool tie[Q];
  bool unres[Q];         // the winner's slot inside leaf bestpos/LEAF is not known yet
  __device__ __forceinline__ void init(float bound_exclusive) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      best[q] = bound_exclusive;
      bestpos[q] = NO_INDEX;
      tie[q] = false;
      unres[q] = false;
    }
  }
  __device__ __forceinline__ void seed(int q, float d, uint32_t pos) {
    if (d < best[q]) {
      best[q] = d;
      bestpos[q] = pos;
    }
  }
  __device__ __forceinline__ float worst(int q) const { return best[q]; }
  __device__ __forceinline__ void leaf(const float* l, uint32_t leaf_id, const float* qx, const float* qy,
                                       const float* qz) {
    float m[Q];
    v2f qx2[Q], qy2[Q], qz2[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      m[q] = __builtin_inff();
      qx2[q] = v2f{qx[q], qx[q]};
      qy2[q] = v2f{qy[q], qy[q]};
      qz2[q] = v2f{qz[q], qz[q]};
    }
#pragma unroll
    for (int j = 0; j < LEAF / 2; ++j) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {  // one broadcast read of the pair serves all Q queries
        const v2f r = pair_dist(l, j, qx2[q], qy2[q], qz2[q]);
        m[q] = __builtin_fminf(m[q], __builtin_fminf(r.x, r.y));
      }
    }
    const uint32_t first = leaf_id * LEAF;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const bool imp = m[q] < best[q];
      // equal minimum in another leaf than the current winner's: index tie, the exact policy decides;
      // equal minimum in the winner's own leaf (a seed re-found -- or another slot at exactly the seed's
      // distance): resolve() looks at the slots
      const bool eq = (m[q] == best[q]) && (bestpos[q] != NO_INDEX);
      const bool same = bestpos[q] / LEAF == leaf_id;
      tie[q] = imp ? false : (tie[q] || (eq && !same));
      const bool look = imp || (eq && same);
      best[q] = imp ? m[q] : best[q];
      bestpos[q] = look ? first : bestpos[q];
      unres[q] = unres[q] || look;
    }
  }
  // Lane-sparse evaluation (traverse(): SPARSE): every lane evaluates ITS OWN leaf (id NO_INDEX = none),
  // staged in batch slot `slot` of the transposed LDS buffer -- a lane only pays for the leaves its own
  // bound could not exclude instead of for the union over the wave's 64 queries.
  static constexpr bool LANE_SPARSE = (Q == 1);
  static constexpr bool NEEDS_W = false;
  __device__ __forceinline__ void leaf_lane(const float* buf, uint32_t slot, uint32_t leaf_id, const float* qx,
                                            const float* qy, const float* qz) {
    // transposed staging: chunk c of slot s at s[c * 16]
    leaf_at<16>(reinterpret_cast<const float4*>(buf) + slot, leaf_id, qx, qy, qz);
  }
  // the same straight from the index's SoA copy (chunk c at s[c]): a leaf that is not staged
  __device__ __forceinline__ void leaf_global(const float* soa, uint32_t leaf_id, const float* qx, const float* qy,
                                              const float* qz) {
    leaf_at<1>(reinterpret_cast<const float4*>(soa + size_t(leaf_id != NO_INDEX ? leaf_id : 0u) * (4 * LEAF)), leaf_id, qx,
               qy, qz);
  }
  template <int STRIDE>
  __device__ __forceinline__ void leaf_at(const float4* s, uint32_t leaf_id, const float* qx, const float* qy,
                                          const float* qz) {
    if (leaf_id != NO_INDEX) {
      const v2f qx2 = {qx[0], qx[0]}, qy2 = {qy[0], qy[0]}, qz2 = {qz[0], qz[0]};
      float m = __builtin_inff();
#pragma unroll
      for (int c4 = 0; c4 < LEAF / 4; ++c4) {
        const float4 X = s[c4 * STRIDE], Y = s[(LEAF / 4 + c4) * STRIDE], Z = s[(2 * (LEAF / 4) + c4) * STRIDE];
        {
          const v2f dx = qx2 - v2f{X.x, X.y}, dy = qy2 - v2f{Y.x, Y.y}, dz = qz2 - v2f{Z.x, Z.y};
          v2f r = dx * dx;
          r = r + dy * dy;
          r = r + dz * dz;
          m = __builtin_fminf(m, __builtin_fminf(r.x, r.y));
        }
        {
          const v2f dx = qx2 - v2f{X.z, X.w}, dy = qy2 - v2f{Y.z, Y.w}, dz = qz2 - v2f{Z.z, Z.w};
          v2f r = dx * dx;
          r = r + dy * dy;
          r = r + dz * dz;
          m = __builtin_fminf(m, __builtin_fminf(r.x, r.y));
        }
      }
      const bool imp = m < best[0];
      const bool eq = (m == best[0]) && (bestpos[0] != NO_INDEX);  // see leaf()
      const bool same = bestpos[0] / LEAF == leaf_id;
      tie[0] = imp ? false : (tie[0] || (eq && !same));
      const bool look = imp || (eq && same);
      best[0] = imp ? m : best[0];
      bestpos[0] = look ? leaf_id * LEAF : bestpos[0];
      unres[0] = unres[0] || look;
    }
  }
  // after the traversal: find the winning slot inside the winning leaf (same arithmetic, same bits)
  __device__ __forceinline__ void resolve(const IndexView& ix, const float* qx, const float* qy, const float* qz) {
    resolve(ix.soa, qx, qy, qz);
  }
  __device__ __forceinline__ void resolve(const float* __restrict__ soa, const float* qx, const float* qy, const float* qz) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (__builtin_amdgcn_ballot_w64(unres[q]) == 0) continue;
      if (unres[q]) {
        const float4* s = reinterpret_cast<const float4*>(soa + size_t(bestpos[q] / LEAF) * LEAF_FLOATS);
        uint32_t hit = 0;
        // the 16 distances on packed v_pk_* math, exactly as the evaluation computed them (scalar: +0.7 % per converged iteration)
        const v2f qx2 = {qx[q], qx[q]}, qy2 = {qy[q], qy[q]}, qz2 = {qz[q], qz[q]};
#pragma unroll
        for (int c4 = 0; c4 < LEAF / 4; ++c4) {
          const float4 X = s[c4], Y = s[LEAF / 4 + c4], Z = s[2 * (LEAF / 4) + c4];
          {
            const v2f dx = qx2 - v2f{X.x, X.y}, dy = qy2 - v2f{Y.x, Y.y}, dz = qz2 - v2f{Z.x, Z.y};
            v2f r = dx * dx;
            r = r + dy * dy;
            r = r + dz * dz;
            hit |= (r.x == best[q] ? 1u : 0u) << (4 * c4);
            hit |= (r.y == best[q] ? 1u : 0u) << (4 * c4 + 1);
          }
          {
            const v2f dx = qx2 - v2f{X.z, X.w}, dy = qy2 - v2f{Y.z, Y.w}, dz = qz2 - v2f{Z.z, Z.w};
            v2f r = dx * dx;
            r = r + dy * dy;
            r = r + dz * dz;
            hit |= (r.x == best[q] ? 1u : 0u) << (4 * c4 + 2);
            hit |= (r.y == best[q] ? 1u : 0u) << (4 * c4 + 3);
          }
        }
        // hit == 0 cannot happen (identical operations); if it ever did, the exact policy takes over
        tie[q] = tie[q] || hit == 0u || (hit & (hit - 1u)) != 0u;
        bestpos[q] = hit ? (bestpos[q] / LEAF) * LEAF + uint32_t(__builtin_ctz(hit)) : NO_INDEX;
        unres[q] = false;
      }
    }
  }
};
typedef NN1MinT<1> NN1Min;

// top-K in registers: ascending (distance, original index) keys + the sorted position of each.
template <int K>
struct TopKReg {
  uint64_t keys[K];
  uint32_t pos[K];
  __device__ __forceinline__ void init(uint64_t k0) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
      keys[i] = k0;
      pos[i] = NO_INDEX;
    }
  }
  static constexpr int QPL = 1;
  __device__ __forceinline__ float worst(int) const { return key_dist(keys[K - 1]); }
  __device__ __forceinline__ void insert(uint64_t k, uint32_t p) {
    // slot j receives old[j-1] if k < old[j-1], else k if k < old[j], else keeps old[j]
    bool below = k < keys[K - 1];  // k < old[j]
#pragma unroll
    for (int j = K - 1; j > 0; --j) {
      const bool up = k < keys[j - 1];
      keys[j] = up ? keys[j - 1] : (below ? k : keys[j]);
      pos[j] = up ? pos[j - 1] : (below ? p : pos[j]);
      below = up;
    }
    keys[0] = below ? k : keys[0];
    pos[0] = below ? p : pos[0];
  }
  // A candidate pair is looked at further only if some lane has d <= its current k-th distance; the
  // exact (distance, index) order is decided inside insert(), so ties stay exact.
  __device__ __forceinline__ void leaf(const float* l, uint32_t leaf_id, const float* qx, const float* qy,
                                       const float* qz) {
    const uint32_t base = leaf_id * LEAF;
    const v2f qx2 = {qx[0], qx[0]}, qy2 = {qy[0], qy[0]}, qz2 = {qz[0], qz[0]};
#pragma unroll
    for (int j = 0; j < LEAF / 2; ++j) {
      const v2f r = pair_dist(l, j, qx2, qy2, qz2);
      const float wd = key_dist(keys[K - 1]);
      const bool e0 = r.x <= wd, e1 = r.y <= wd;
      if (__builtin_amdgcn_ballot_w64(e0 || e1) != 0) {  // wave-uniform branch
        if (__builtin_amdgcn_ballot_w64(e0) != 0)
          insert(make_key(r.x, __float_as_uint(l[3 * LEAF + 2 * j])), base + 2 * j);
        if (__builtin_amdgcn_ballot_w64(r.y <= key_dist(keys[K - 1])) != 0)
          insert(make_key(r.y, __float_as_uint(l[3 * LEAF + 2 * j + 1])), base + 2 * j + 1);
      }
    }
  }
  // lane-sparse evaluation from the transposed LDS staging buffer (see traverse(): SPARSE): the lane's
  // own leaf; a candidate goes through the insertion network only if some active lane still wants it.
  // (Per-lane qualifier masks + an extraction loop -- each lane inserting only ITS qualifying candidates -- were
  // measured: the k-th distance is loose for most of the traversal, the loop ran ~10 times per round instead of
  // the hoped-for 6, and the extra registers cost a wave per SIMD: 6.2 -> 8.1 ms for k = 8 normals of 10M points.)
  static constexpr bool LANE_SPARSE = true;
  static constexpr bool NEEDS_W = true;
  __device__ __forceinline__ void leaf_lane(const float* buf, uint32_t slot, uint32_t leaf_id, const float* qx,
                                            const float* qy, const float* qz) {
    if (leaf_id != NO_INDEX) {
      PCLHIP_LANE_MASKED_REGION;  // the ballots below run among the lanes that have a leaf
      const float4* s = reinterpret_cast<const float4*>(buf) + slot;  // chunk c at s[c * 16]
      const uint32_t base = leaf_id * LEAF;
      const v2f qx2 = {qx[0], qx[0]}, qy2 = {qy[0], qy[0]}, qz2 = {qz[0], qz[0]};
#pragma unroll
      for (int c4 = 0; c4 < LEAF / 4; ++c4) {
        const float4 X = s[c4 * 16], Y = s[(4 + c4) * 16], Z = s[(8 + c4) * 16], W = s[(12 + c4) * 16];
        v2f r0, r1;
        {
          const v2f dx = qx2 - v2f{X.x, X.y}, dy = qy2 - v2f{Y.x, Y.y}, dz = qz2 - v2f{Z.x, Z.y};
          r0 = dx * dx;
          r0 = r0 + dy * dy;
          r0 = r0 + dz * dz;
        }
        {
          const v2f dx = qx2 - v2f{X.z, X.w}, dy = qy2 - v2f{Y.z, Y.w}, dz = qz2 - v2f{Z.z, Z.w};
          r1 = dx * dx;
          r1 = r1 + dy * dy;
          r1 = r1 + dz * dz;
        }
        const float ds[4] = {r0.x, r0.y, r1.x, r1.y}, ws[4] = {W.x, W.y, W.z, W.w};
        const float m = __builtin_fminf(__builtin_fminf(ds[0], ds[1]), __builtin_fminf(ds[2], ds[3]));
        if (__builtin_amdgcn_ballot_w64(m <= key_dist(keys[K - 1])) == 0) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (__builtin_amdgcn_ballot_w64(ds[j] <= key_dist(keys[K - 1])) != 0)
            insert(make_key(ds[j], __float_as_uint(ws[j])), base + uint32_t(4 * c4 + j));
        }
      }
    }
  }
};

// The k smallest DISTANCES only, sorted ascending.  Without identities to carry, inserting into a sorted run is
// one v_med3_f32 per slot (new d[j] = median(d[j-1], x, d[j])) instead of the compare + select chains of 64-bit
// (distance, index) keys with their positions: 8 instructions per candidate for k = 8 where TopKReg needs ~72.
// Used as the first of two passes (see normals_kernel): it yields the exact k-th distance; CollectLE then
// gathers the candidates up to that distance.
// Every leaf that MAY hold one of the final k neighbours is remembered per lane (`rec`, entry r of this lane at
// rec[r * 64] in LDS): a leaf is recorded when its nearest point is not beyond the lane's k-th distance of that moment.
// The final k-th distance is never larger, so a point within it -- one that ties with it included -- lies in a recorded
// leaf, and the second pass reads those leaves again instead of walking the index a second time.
// (The lane's own leaf, evaluated before the traversal, is not in the list: the caller knows it.)
#ifndef PCLHIP_REC_CAP
#define PCLHIP_REC_CAP 7  // (a build with 1 drives most waves through the fallback: scripts/final_evidence.sh tests it)
#endif
constexpr uint32_t REC_CAP = PCLHIP_REC_CAP;  // recorded leaves per lane; a lane that needs more sends its wave through the traversal
constexpr uint32_t REC_MIN_ROW = 8;                          // the ids' rows fill 2 KB: the list of that traversal
#ifndef PCLHIP_NRM_WAVES
#define PCLHIP_NRM_WAVES 4  // waves per SIMD of normals_kernel<8>: 4 = box-only LDS layout, records without their minima
#endif
constexpr bool REC_MINS = PCLHIP_NRM_WAVES < 4;  // keep every record's nearest distance (filters the second pass)
constexpr uint32_t REC_BYTES = (REC_MIN_ROW + (REC_MINS ? REC_CAP : 0u)) * WAVE * 4;  // per wave
template <int K>
struct TopKDist {
  float d[K];
  uint32_t* rec;   // LDS, this lane's column (nullptr: nothing is recorded): ids in rows [0, REC_CAP), the leaves'
                   // smallest distances in rows [REC_MIN_ROW, REC_MIN_ROW + REC_CAP)
  uint32_t nrec;   // leaves this lane would have recorded (> REC_CAP: the list is incomplete)
  __device__ __forceinline__ void init(uint32_t* rec_column = nullptr) {
#pragma unroll
    for (int i = 0; i < K; ++i) d[i] = __builtin_inff();
    rec = rec_column;
    nrec = 0;
  }
  __device__ __forceinline__ void record(uint32_t leaf_id, float nearest) {
    if (nrec < REC_CAP) {
      rec[nrec * WAVE] = leaf_id;
      if constexpr (REC_MINS) rec[(REC_MIN_ROW + nrec) * WAVE] = __float_as_uint(nearest);
    }
    ++nrec;
  }
  static constexpr int QPL = 1;
  static constexpr bool LANE_SPARSE = true;
  static constexpr bool NEEDS_W = false;
  __device__ __forceinline__ float worst(int) const { return d[K - 1]; }
  __device__ __forceinline__ void insert(float x) {
#pragma unroll
    for (int j = K - 1; j > 0; --j) d[j] = __builtin_amdgcn_fmed3f(d[j - 1], x, d[j]);
    d[0] = __builtin_fminf(d[0], x);
  }
  // the 16 candidates of one leaf block: chunk c (16 bytes) at s[c * STRIDE]
  // returns the smallest of the 16 distances
  template <int STRIDE>
  __device__ __forceinline__ float block(const float4* s, const float* qx, const float* qy, const float* qz) {
    const v2f qx2 = {qx[0], qx[0]}, qy2 = {qy[0], qy[0]}, qz2 = {qz[0], qz[0]};
    float m = __builtin_inff();
#pragma unroll
    for (int c4 = 0; c4 < LEAF / 4; ++c4) {
      const float4 X = s[c4 * STRIDE], Y = s[(4 + c4) * STRIDE], Z = s[(8 + c4) * STRIDE];
      v2f r0, r1;
      {
        const v2f dx = qx2 - v2f{X.x, X.y}, dy = qy2 - v2f{Y.x, Y.y}, dz = qz2 - v2f{Z.x, Z.y};
        r0 = dx * dx;
        r0 = r0 + dy * dy;
        r0 = r0 + dz * dz;
      }
      {
        const v2f dx = qx2 - v2f{X.z, X.w}, dy = qy2 - v2f{Y.z, Y.w}, dz = qz2 - v2f{Z.z, Z.w};
        r1 = dx * dx;
        r1 = r1 + dy * dy;
        r1 = r1 + dz * dz;
      }
      const float m4 = __builtin_fminf(__builtin_fminf(r0.x, r0.y), __builtin_fminf(r1.x, r1.y));
      m = __builtin_fminf(m, m4);
      // (skipping the four insertions when no active lane's list can change -- one ballot per chunk -- was measured:
      // 2.12 against 2.08 ms, the branch costs more than the medians it saves)
      insert(r0.x);
      insert(r0.y);
      insert(r1.x);
      insert(r1.y);
    }
    return m;
  }
  __device__ __forceinline__ void leaf_lane(const float* buf, uint32_t slot, uint32_t leaf_id, const float* qx,
                                            const float* qy, const float* qz) {
    if (leaf_id != NO_INDEX) {
      const float before = d[K - 1];
      const float m = block<16>(reinterpret_cast<const float4*>(buf) + slot, qx, qy, qz);  // transposed staging
      if (rec != nullptr && m <= before) record(leaf_id, m);
    }
  }
  // Self-queries: the lane's OWN leaf is evaluated before the traversal (straight from the SoA copy), so the search
  // starts with a k-th distance of a few point spacings -- tight mode, below the root -- instead of +inf; the
  // traversal must then not hand that leaf to this lane again (a distance would be inserted twice).
  uint32_t exclude = NO_INDEX;
  __device__ __forceinline__ void seed_own_leaf(const float* soa, uint32_t leaf_id, const float* qx, const float* qy,
                                                const float* qz) {
    block<1>(reinterpret_cast<const float4*>(soa + size_t(leaf_id) * LEAF_FLOATS), qx, qy, qz);
    exclude = leaf_id;
  }
};

// Second pass: sorted positions of every candidate whose distance is <= thr (finite), appended to a per-lane
// list in LDS (entry c of this lane at list[c * 64]); more than K of them (exact distance ties at the k-th
// distance) raise `over`, and the caller lets the exact (distance, index) policy decide for that lane.
template <int K>
struct CollectLE {
  float thr;
  uint32_t cnt;
  bool over;
  uint32_t* list;  // LDS, this lane's column
  static constexpr int QPL = 1;
  static constexpr bool LANE_SPARSE = true;
  static constexpr bool NEEDS_W = false;
  __device__ __forceinline__ float worst(int) const { return thr; }
  __device__ __forceinline__ void take(float d, uint32_t p) {
    if (d <= thr && d < __builtin_inff()) {
      if (cnt < uint32_t(K)) list[cnt * WAVE] = p;
      else over = true;
      ++cnt;
    }
  }
  __device__ __forceinline__ void leaf_lane(const float* buf, uint32_t slot, uint32_t leaf_id, const float* qx,
                                            const float* qy, const float* qz) {
    leaf_at<16>(reinterpret_cast<const float4*>(buf) + slot, leaf_id, qx, qy, qz);  // transposed staging
  }
  // the same straight from the index's SoA copy: the lane's own leaf, not one of a staged batch
  __device__ __forceinline__ void leaf_global(const float* soa, uint32_t leaf_id, const float* qx, const float* qy,
                                              const float* qz) {
    leaf_at<1>(reinterpret_cast<const float4*>(soa + size_t(leaf_id != NO_INDEX ? leaf_id : 0u) * LEAF_FLOATS), leaf_id, qx,
               qy, qz);
  }
  template <int STRIDE>
  __device__ __forceinline__ void leaf_at(const float4* s, uint32_t leaf_id, const float* qx, const float* qy,
                                          const float* qz) {
    if (leaf_id != NO_INDEX) {
      const uint32_t base = leaf_id * LEAF;
      const v2f qx2 = {qx[0], qx[0]}, qy2 = {qy[0], qy[0]}, qz2 = {qz[0], qz[0]};
#pragma unroll
      for (int c4 = 0; c4 < LEAF / 4; ++c4) {
        const float4 X = s[c4 * STRIDE], Y = s[(4 + c4) * STRIDE], Z = s[(8 + c4) * STRIDE];
        v2f r0, r1;
        {
          const v2f dx = qx2 - v2f{X.x, X.y}, dy = qy2 - v2f{Y.x, Y.y}, dz = qz2 - v2f{Z.x, Z.y};
          r0 = dx * dx;
          r0 = r0 + dy * dy;
          r0 = r0 + dz * dz;
        }
        {
          const v2f dx = qx2 - v2f{X.z, X.w}, dy = qy2 - v2f{Y.z, Y.w}, dz = qz2 - v2f{Z.z, Z.w};
          r1 = dx * dx;
          r1 = r1 + dy * dy;
          r1 = r1 + dz * dz;
        }
        take(r0.x, base + uint32_t(4 * c4));
        take(r0.y, base + uint32_t(4 * c4 + 1));
        take(r1.x, base + uint32_t(4 * c4 + 2));
        take(r1.y, base + uint32_t(4 * c4 + 3));
      }
    }
  }
};

// The second pass without a walk: the lane reads ONE leaf (its own, or one it recorded in the first pass) straight from
// the index's SoA copy and appends every candidate up to thr (finite) as an 8-bit code -- (row of the record << 4) | slot
// in the leaf -- to a 64-bit register: eight codes, more candidates than that are only counted.  Same distance arithmetic
// as every other policy; "d <= thr" is read off the sign of thr - d (a difference of floats is negative exactly when
// d > thr), sixteen of them shifted into one mask.
struct CollectCodes {
  float thr;
  uint32_t cnt;
  uint64_t codes;
  __device__ __forceinline__ void leaf(const float* soa, uint32_t leaf_id, uint32_t row, const float* qx, const float* qy,
                                       const float* qz) {
    uint32_t hits = 0;  // bit 15 - j: point j is a candidate
    if (leaf_id != NO_INDEX) {
      const float4* s = reinterpret_cast<const float4*>(soa + size_t(leaf_id) * LEAF_FLOATS);
      const v2f qx2 = {qx[0], qx[0]}, qy2 = {qy[0], qy[0]}, qz2 = {qz[0], qz[0]}, t2 = {thr, thr};
      uint32_t over = 0;
#pragma unroll
      for (int c4 = 0; c4 < LEAF / 4; ++c4) {
        const float4 X = s[c4], Y = s[4 + c4], Z = s[8 + c4];
        v2f r0, r1;
        {
          const v2f dx = qx2 - v2f{X.x, X.y}, dy = qy2 - v2f{Y.x, Y.y}, dz = qz2 - v2f{Z.x, Z.y};
          r0 = dx * dx;
          r0 = r0 + dy * dy;
          r0 = r0 + dz * dz;
        }
        {
          const v2f dx = qx2 - v2f{X.z, X.w}, dy = qy2 - v2f{Y.z, Y.w}, dz = qz2 - v2f{Z.z, Z.w};
          r1 = dx * dx;
          r1 = r1 + dy * dy;
          r1 = r1 + dz * dz;
        }
        const v2f u0 = t2 - r0, u1 = t2 - r1;
        over = __builtin_amdgcn_alignbit(over, __float_as_uint(u0.x), 31);  // (over << 1) | sign
        over = __builtin_amdgcn_alignbit(over, __float_as_uint(u0.y), 31);
        over = __builtin_amdgcn_alignbit(over, __float_as_uint(u1.x), 31);
        over = __builtin_amdgcn_alignbit(over, __float_as_uint(u1.y), 31);
      }
      hits = ~over & 0xFFFFu;
    }
    while (__builtin_amdgcn_ballot_w64(hits != 0) != 0) {
      if (hits != 0) {
        const uint32_t b = 31u - uint32_t(__builtin_clz(hits));  // ascending slots
        const uint64_t code = uint64_t((row << 4) | (15u - b));
        if (cnt < 8u) codes |= code << (8u * cnt);
        ++cnt;
        hits &= ~(1u << b);
      }
    }
  }
};

// top-k for arbitrary k in a per-lane binary max-heap in global memory, layout heap[slot*nq + q]
// (slot-major so that lanes touching the same slot coalesce).
struct TopKHeap {
  uint64_t* heap;  // + q already applied
  size_t stride;   // nq
  int k;
  uint64_t root;   // cached heap[0] (current worst)
  __device__ __forceinline__ void init(uint64_t k0) {
    for (int i = 0; i < k; ++i) heap[size_t(i) * stride] = k0;
    root = k0;
  }
  static constexpr int QPL = 1;
  __device__ __forceinline__ float worst(int) const { return key_dist(root); }
  __device__ void replace_root(uint64_t key) {
    int i = 0;
    for (;;) {
      int l = 2 * i + 1, r = l + 1, big = i;
      uint64_t bv = key;
      if (l < k) {
        const uint64_t lv = heap[size_t(l) * stride];
        if (lv > bv) {
          bv = lv;
          big = l;
        }
      }
      if (r < k) {
        const uint64_t rv = heap[size_t(r) * stride];
        if (rv > bv) {
          bv = rv;
          big = r;
        }
      }
      if (big == i) break;
      heap[size_t(i) * stride] = bv;
      i = big;
    }
    heap[size_t(i) * stride] = key;
    root = heap[0];
  }
  __device__ __forceinline__ void leaf(const float* l, uint32_t leaf_id, const float* qx, const float* qy,
                                       const float* qz) {
    for (int c = 0; c < LEAF; ++c) {
      const float d = l2_simple(qx[0], qy[0], qz[0], l[c], l[LEAF + c], l[2 * LEAF + c]);
      const uint64_t key = make_key(d, __float_as_uint(l[3 * LEAF + c]));
      if (key < root) replace_root(key);
    }
  }
  // in-place heap sort -> ascending keys in heap[0..k)
  __device__ void sort_ascending() {
    for (int end = k - 1; end > 0; --end) {
      const uint64_t top = heap[0];
      const uint64_t last = heap[size_t(end) * stride];
      heap[size_t(end) * stride] = top;
      int i = 0;
      for (;;) {
        int l = 2 * i + 1, r = l + 1, big = i;
        uint64_t bv = last;
        if (l < end) {
          const uint64_t lv = heap[size_t(l) * stride];
          if (lv > bv) {
            bv = lv;
            big = l;
          }
        }
        if (r < end) {
          const uint64_t rv = heap[size_t(r) * stride];
          if (rv > bv) {
            bv = rv;
            big = r;
          }
        }
        if (big == i) break;
        heap[size_t(i) * stride] = bv;
        i = big;
      }
      heap[size_t(i) * stride] = last;
    }
  }
};

// Optional work counters (wave-uniform, live in SGPRs; flushed by the kernels when the context asked
// for statistics).  [0] interior nodes scanned, [1] leaves that passed the group test, [2] leaves
// that passed the per-lane test (= 16-candidate all-pairs blocks), [3] stack pushes, [4] groups; the stand-off path
// (standoff.hpp) adds [5] groups it finished, [6] leaves it collected, [7] union slots after the row cull.
struct TraverseStats {
  uint32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// -DPCLHIP_VERIFY_BOUNDS (scripts/build_variant.sh verify "-DPCLHIP_VERIFY_BOUNDS"; tests/test_gpu_verify_bounds.py): the
// two bounds that are NOT bit-monotone -- the per-lane disc bound (point_disc_lb / so_disc_lb2) and the reach filters
// (row_reach_alive / so_reach_alive) -- are checked where they cull.  A cull claims "no point of this leaf is within the
// lane's bound of this moment"; the lane computes the leaf's true minimum distance (l2_simple on its 16 points) and counts
// the claim in slot 6 and a broken claim (minimum <= bound) in slot 7 of the work counters (per lane, summed by
// flush_stats).  Nothing of this exists in the default build.
#ifdef PCLHIP_VERIFY_BOUNDS
__device__ __forceinline__ float l2_simple(float qx, float qy, float qz, float cx, float cy, float cz);
static __device__ __noinline__ void verify_culled_leaf(const float* __restrict__ soa, uint32_t id, float qx, float qy, float qz, float worst,
                                                bool active, TraverseStats& ts) {
  if (active && id != NO_INDEX) {
    const float* l = soa + size_t(id) * (4 * LEAF);
    float m = __builtin_inff();
    for (int c = 0; c < LEAF; ++c) m = fminf(m, l2_simple(qx, qy, qz, l[c], l[LEAF + c], l[2 * LEAF + c]));
    ++ts.c[6];
    if (m <= worst) ++ts.c[7];
  }
}
#define PCLHIP_VERIFY_CULL(soa, id, qx, qy, qz, worst, active, ts) verify_culled_leaf(soa, id, qx, qy, qz, worst, active, ts)
#else
#define PCLHIP_VERIFY_CULL(soa, id, qx, qy, qz, worst, active, ts) (void)0
#endif

// does the policy offer the lane-sparse leaf evaluation?
template <class P, class = void>
struct lane_sparse_of { static constexpr bool value = false; };
template <class P>
struct lane_sparse_of<P, decltype(void(P::LANE_SPARSE))> { static constexpr bool value = P::LANE_SPARSE; };

// does this lane's policy still want leaf `id`?  (policies with an `exclude` member have evaluated it already)
template <class P, class = void>
struct has_exclude { static constexpr bool value = false; };
template <class P>
struct has_exclude<P, decltype(void(P::exclude))> { static constexpr bool value = true; };
template <class P>
__device__ __forceinline__ bool pol_wants(const P& pol, uint32_t id) {
  if constexpr (has_exclude<P>::value) return id != pol.exclude;
  else return true;
}

// ---- the traversal --------------------------------------------------------------------------------
// `wl` is this wave's LDS working set.  Must be called by all 64 lanes.
// Per-block LDS copy of the boxes of the top tree levels (IndexView::topcache): filled once per block.
__device__ __forceinline__ void load_top_cache(const IndexView& ix, Box* topbox) {
  for (uint32_t i = threadIdx.x; i < ix.cache_count; i += blockDim.x) topbox[i] = ix.topcache[i];
  __syncthreads();
}

// per-lane maximum of the policy's bounds over the lane's valid queries
template <class Policy>
__device__ __forceinline__ float lane_worst(const Policy& pol, const bool* valid) {
  float w = 0.0f;
#pragma unroll
  for (int q = 0; q < Policy::QPL; ++q) w = fmaxf(w, valid[q] ? pol.worst(q) : 0.0f);
  return w;
}

// qx/qy/qz/valid: Policy::QPL queries per lane (the wave owns 64*QPL spatially compact queries).
template <class Policy, bool SPARSE = false, class WL = WaveLds>
__device__ __forceinline__ void traverse(const IndexView& ix, const float* qx, const float* qy, const float* qz,
                                         const bool* valid, Policy& pol, WL& wl, const Box* topbox,
                                         TraverseStats& ts, uint32_t start_leaf = NO_INDEX, bool allow_disc = false) {
  constexpr int QPL = Policy::QPL;
  const int lane = threadIdx.x & (WAVE - 1);
  bool any_valid = false;
#pragma unroll
  for (int q = 0; q < QPL; ++q) any_valid = any_valid || valid[q];
  if (__builtin_amdgcn_ballot_w64(any_valid) == 0 || ix.n == 0) return;
  ++ts.c[4];
  const float BIG = 3.402823466e+38f;
  const float INF = __builtin_inff();
  // bounding box of the wave's queries (per-lane fold over the lane's queries, then one reduction)
  float lx0 = BIG, ly0 = BIG, lz0 = BIG, hx0 = -BIG, hy0 = -BIG, hz0 = -BIG;
#pragma unroll
  for (int q = 0; q < QPL; ++q) {
    if (valid[q]) {
      lx0 = fminf(lx0, qx[q]); ly0 = fminf(ly0, qy[q]); lz0 = fminf(lz0, qz[q]);
      hx0 = fmaxf(hx0, qx[q]); hy0 = fmaxf(hy0, qy[q]); hz0 = fmaxf(hz0, qz[q]);
    }
  }
  float T = lane_worst(pol, valid);  // wave pruning radius (squared)
  wave_min3_max4(lx0, ly0, lz0, hx0, hy0, hz0, T);
  const float Qlx = lx0, Qly = ly0, Qlz = lz0, Qhx = hx0, Qhy = hy0, Qhz = hz0;
  const float gdiag2 = (Qhx - Qlx) * (Qhx - Qlx) + (Qhy - Qly) * (Qhy - Qly) + (Qhz - Qlz) * (Qhz - Qlz);
  // Disc bounds pay where queries STAND OFF the indexed surface (the unseeded first iteration of a
  // registration): the caller says so.  Self-queries (normals) and seeded queries sit on or near the surface,
  // inside the discs of all the leaves around them, where a disc excludes nothing a box does not.
  constexpr int LS = WL::LIST_STRIDE;  // 3: (box | disc centre, id) (box, bound) (disc normal); 2: boxes only
  const bool have_disc = LS == 3 && allow_disc && ix.disc != nullptr;
  float* const rad = wl.radii();
  uint2* const stack = wl.stack;

  int sp = 0;
  uint32_t level = uint32_t(ix.top) + 1u, node = 0u;  // virtual root above the top level
  // Seeded searches need not start at the root.  The leaves are the cells of a kd partition (every aligned
  // block of 16 * 4^j points is one cell, index_build.hip), so cells of different nodes have disjoint
  // interiors and a node's box lies inside its cell.  If the box of the wave's queries, grown by the wave
  // radius, lies STRICTLY inside the box of the level-2 (64 leaves) or level-3 (4096 leaves) node that holds
  // the hint leaf, every point within any lane's bound (equal distances included) belongs to that node:
  // the descent starts there.  The margin covers the rounding of the differences and of the squares.
  if (start_leaf != NO_INDEX && T < INF) {
    const auto inside = [&](const Box& b) {
      const float d = fminf(fminf(fminf(Qlx - b.lo.x, b.hi.x - Qhx), fminf(Qly - b.lo.y, b.hi.y - Qhy)),
                            fminf(Qlz - b.lo.z, b.hi.z - Qhz));
      return d > 0.0f && d * d * 0.999999f > T;
    };
    const bool has2 = ix.top >= 2, has3 = ix.top >= 3;
    Box b2 = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)}, b3 = b2;
    // (the node's kd cell where the index carries cells: the box lies inside it, see IndexView::cell2)
    if (has2) b2 = ix.cell2 != nullptr ? ix.cell2[start_leaf >> 6] : ix.box[2][start_leaf >> 6];   // wave-uniform addresses: both loads are in flight together
    if (has3) b3 = ix.cell3 != nullptr ? ix.cell3[start_leaf >> 12] : ix.box[3][start_leaf >> 12];
    if (has2 && inside(b2)) {
      level = 2u;
      node = start_leaf >> 6;
    } else if (has3 && inside(b3)) {
      level = 3u;
      node = start_leaf >> 12;
    }
  }
  bool have = true;
  for (;;) {
    if (!have) {
      if (sp == 0) break;
      --sp;
      __builtin_amdgcn_wave_barrier();
      const uint2 e = stack[sp];
      const uint32_t ex = uniform_u32(e.x), ey = uniform_u32(e.y);
      if (__uint_as_float(ey) > T) continue;  // pruned since it was pushed
      level = ex >> 28;
      node = ex & 0x0FFFFFFFu;
    }
    have = false;
    ++ts.c[0];
    const uint32_t cl = level - 1u;  // level of the children
    const uint32_t first = node * FANOUT;
    // level table entry: a wave-uniform select over kernel arguments (SGPRs) -- no memory access on
    // the traversal's critical path
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
    float lx = 0, ly = 0, lz = 0, hx = 0, hy = 0, hz = 0;
    if (int(cl) >= ix.cache_from) {  // upper levels: boxes come from the block's LDS copy
      if (has) {
        const Box b = topbox[coff + first + lane];
        lx = b.lo.x; ly = b.lo.y; lz = b.lo.z;
        hx = b.hi.x; hy = b.hi.y; hz = b.hi.z;
      }
    } else if (has) {
      const Box b = level_box[first + lane];
      lx = b.lo.x; ly = b.lo.y; lz = b.lo.z;
      hx = b.hi.x; hy = b.hi.y; hz = b.hi.z;
    }
    float lbG = has ? box_box_lb(Qlx, Qly, Qlz, Qhx, Qhy, Qhz, lx, ly, lz, hx, hy, hz) : INF;
    // Visiting order.  Cold or lukewarm bounds (wave radius not yet small against the group's own
    // extent, e.g. the first ICP iterations): ascending distance to the group box, so the bounds
    // collapse after the first few leaves and everything farther is cut off at once.  Tight bounds
    // (seeded steady state): plain index order, no ranking work.
    // (the factor counted on the emulation at 1M points, second launch of an alignment, per group: 4 -> 21.7 per-lane tests
    // and 10.7 rounds, 8 -> 20.1 / 9.5, 16 -> 19.3 / 9.0, 64 -> 18.9 / 9.0, 1024 -> 18.9 / 9.0 but 7.2 rounds once converged)
    const bool ordered = T * 16.0f > gdiag2;
    // Leaves of a loose search are bounded by their discs (see point_disc_lb) once the queries stand off
    // farther than a few leaf sizes: closer in, a disc excludes nothing a box does not and costs twice the test.
    bool use_disc = false;
    float4 dcR = make_float4(0, 0, 0, 0), dnh = make_float4(0, 0, 0, 0);
    if (have_disc && ordered && cl == 1u && T > ix.disc_from) {  // disc_from: a few mean leaf diagonals, squared
      use_disc = true;
      if (has) {
        dcR = ix.disc[2 * (first + lane)];
        dnh = ix.disc[2 * (first + lane) + 1];
        // (a group-level disc bound -- smallest stand-off over the group against the largest radius -- was here: at a
        // stand-off every slack in it is multiplied by the stand-off, it excluded nothing the box bound does not, and
        // cost 45 instructions per scan; the lanes' own disc tests below are what prunes)
      }
    }
    const bool alive = has && !(lbG > T);
    const uint64_t mask = __builtin_amdgcn_ballot_w64(alive);
    if (mask == 0) continue;
    if (cl == 1u) {
      // ---- rank the surviving leaves and publish (box, id, lbG) in LDS --------------------------
      uint32_t rank = uint32_t(__builtin_popcountll(mask & ((1ull << lane) - 1ull)));
      if (ordered) {
        // Exact rank by UNIQUE 32-bit keys: the (non-negative) bound with its six low mantissa bits replaced by the
        // lane -- one readlane, one compare and one add-with-carry per surviving leaf instead of a two-key
        // comparison.  The list then carries the truncated bound: it is <= the true one, so "every remaining
        // leaf is farther than the wave radius" stays a valid cut.
        const uint32_t key = (__float_as_uint(lbG) & ~63u) | uint32_t(lane);
        rank = 0;
        for (uint64_t m2 = mask; m2; m2 &= m2 - 1) {
          const uint32_t sk = uint32_t(__builtin_amdgcn_readlane(int(key), __builtin_ctzll(m2)));
          rank += sk < key ? 1u : 0u;
        }
        lbG = __uint_as_float(key & ~63u);
      }
      const uint32_t n_alive = uint32_t(__builtin_popcountll(mask));
      if (alive) {
        if (use_disc) {
          wl.list[LS * rank] = make_float4(dcR.x, dcR.y, dcR.z, __uint_as_float(first + uint32_t(lane)));
          wl.list[LS * rank + 1] = make_float4(0.0f, 0.0f, 0.0f, lbG);
          wl.list[LS * rank + 2] = dnh;
          rad[rank] = dcR.w;
        } else {
          wl.list[LS * rank] = make_float4(lx, ly, lz, __uint_as_float(first + uint32_t(lane)));
          wl.list[LS * rank + 1] = make_float4(hx, hy, hz, lbG);
        }
      }
      __builtin_amdgcn_wave_barrier();
      if constexpr (SPARSE) {
        static_assert(QPL == 1 && lane_sparse_of<Policy>::value, "lane-sparse evaluation: one query per lane");
        // Lane-sparse evaluation: a lane only evaluates the leaves ITS OWN bound cannot exclude.
        // The listed leaves are staged in LDS 16 at a time with the LDS-DMA path, TRANSPOSED: chunk c
        // (16 bytes) of the leaf in batch slot s lands at ((c * 16 + s) * 16) bytes, so lanes that read
        // the same chunk of different leaves hit different banks.  The DMA runs under the box tests.
        constexpr int NCHUNK = Policy::NEEDS_W ? 16 : 12;  // x[16] y[16] z[16] (+ w[16]) in 16-byte chunks
        static_assert(NCHUNK * LEAF_BATCH * 4 <= WL::BUF_FLOATS, "staging buffer too small for this policy");
        const auto round = [&](uint32_t slot, uint32_t id) {
          const uint64_t act = __builtin_amdgcn_ballot_w64(id != NO_INDEX);
          if (act == 0) return;
          ++ts.c[2];
#ifdef PCLHIP_STATS_LANES
          ts.c[3] += uint32_t(__builtin_popcountll(act));
#endif
          pol.leaf_lane(wl.buf, slot, id, qx, qy, qz);
        };
        const float before = pol.worst(0);
        const bool loose = ordered;
        bool cut = false;
        // LDS-DMA of the candidate blocks of list entries [b0, b0 + nb) into the (transposed) staging buffer
        const auto stage = [&](uint32_t b0, uint32_t nb) {
          const uint32_t slot = uint32_t(lane) & 15u;
          uint32_t leaf_id = 0;
          if (slot < nb) leaf_id = __float_as_uint(wl.list[LS * (b0 + slot)].w);
#pragma unroll
          for (int i = 0; i < NCHUNK / 4; ++i) {
            if (slot < nb) {
              const float* src = ix.soa + size_t(leaf_id) * LEAF_FLOATS + (i * 4 + (lane >> 4)) * 4;
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                               (__attribute__((address_space(3))) void*)(wl.buf + i * (WAVE * 4)), 16, 0,
                                               0);
            }
          }
        };
        for (uint32_t b0 = 0; b0 < n_alive && !cut; b0 += LEAF_BATCH) {
          const uint32_t nb = (n_alive - b0) < uint32_t(LEAF_BATCH) ? (n_alive - b0) : uint32_t(LEAF_BATCH);
          if (b0 != 0 && use_disc && pair_mode) {
            // ---- pair mode: the rest of a disc list after its first batch --------------------------------------------
            // The first batch (the 16 leaves nearest to the group) has given every lane a radius close to its final one.
            // From here the remaining (lane, leaf) pairs are culled hierarchically instead of one leaf at a time for all
            // 64 lanes: (row of 16 lanes, leaf) pairs by the reach filter, 64 pairs per pass; then every lane tests only
            // the leaves alive for ITS row; then the needed leaves are evaluated, lane-sparse, batch by batch.  Nothing
            // tightens inside this block, which is why it only follows a first batch walked the sequential way (starting
            // later lists of the group in pair mode was measured: 15 -> 24 evaluation rounds, 3.10 -> 3.65 ms).
            const float BIGF = 3.402823466e+38f;
            RowReach rr;
            float ngx = 0.0f, ngy = 0.0f, ngz = 0.0f;
            {
              // the row's queries: bounding box -> centre and radius (rounded up); invalid lanes take no part
              float l0 = valid[0] ? qx[0] : BIGF, l1 = valid[0] ? qy[0] : BIGF, l2 = valid[0] ? qz[0] : BIGF;
              float h0 = valid[0] ? qx[0] : -BIGF, h1 = valid[0] ? qy[0] : -BIGF, h2 = valid[0] ? qz[0] : -BIGF;
              row_min3_f(l0, l1, l2);
              row_max3_f(h0, h1, h2);
              const bool any = !(l0 > h0);
              rr.cx = any ? 0.5f * (l0 + h0) : 0.0f;
              rr.cy = any ? 0.5f * (l1 + h1) : 0.0f;
              rr.cz = any ? 0.5f * (l2 + h2) : 0.0f;
              const float ex = any ? h0 - l0 : 0.0f, ey = any ? h1 - l1 : 0.0f, ez = any ? h2 - l2 : 0.0f;
              rr.rS = __fsqrt_rn((ex * ex + ey * ey) + ez * ez) * 0.5000005f +
                      1e-6f * ((fabsf(rr.cx) + fabsf(rr.cy)) + fabsf(rr.cz));
              // group direction: the four rows are the quadrants of the 64-query patch (kd order), the cross product of
              // the two diagonals of their centres is its normal.  Any direction is valid; a poor one only filters less.
              const float r0x = readlane_f(rr.cx, 0), r0y = readlane_f(rr.cy, 0), r0z = readlane_f(rr.cz, 0);
              const float r1x = readlane_f(rr.cx, 16), r1y = readlane_f(rr.cy, 16), r1z = readlane_f(rr.cz, 16);
              const float r2x = readlane_f(rr.cx, 32), r2y = readlane_f(rr.cy, 32), r2z = readlane_f(rr.cz, 32);
              const float r3x = readlane_f(rr.cx, 48), r3y = readlane_f(rr.cy, 48), r3z = readlane_f(rr.cz, 48);
              const float ux = r3x - r0x, uy = r3y - r0y, uz = r3z - r0z, vx = r2x - r1x, vy = r2y - r1y, vz = r2z - r1z;
              const float cxn = uy * vz - uz * vy, cyn = uz * vx - ux * vz, czn = ux * vy - uy * vx;
              const float l2n = cxn * cxn + cyn * cyn + czn * czn;
              if (l2n > 1e-30f && l2n < 1e30f) {
                const float il = __frsqrt_rn(l2n);
                ngx = cxn * il;
                ngy = cyn * il;
                ngz = czn * il;
              }
              const float a_own = (ngx * (qx[0] - rr.cx) + ngy * (qy[0] - rr.cy)) + ngz * (qz[0] - rr.cz);
              const float rho = __fsqrt_rn(valid[0] ? pol.worst(0) : 0.0f) * 1.000004f;
              rr.Up = valid[0] ? rho - a_own : -BIGF;
              rr.Um = valid[0] ? rho + a_own : -BIGF;
              rr.rho = valid[0] ? rho : 0.0f;
              row_max3_f(rr.Up, rr.Um, rr.rho);
            }
            // (row, leaf) pairs: lane (r, s) tests entry e0 + s for row r; a row keeps its 16 bits of every ballot
            const uint32_t sub = uint32_t(lane) & 15u, rshift = (uint32_t(lane) >> 4) * 16u;
            uint64_t rowmask = 0;
            for (uint32_t e0 = b0; e0 < n_alive; e0 += 16u) {
              const uint32_t e = e0 + sub;
              bool al = false;
              if (e < n_alive) {
                const float4 ea = wl.list[LS * e], eb = wl.list[LS * e + 1], es = wl.list[LS * e + 2];
                al = !(eb.w > T) && (!(rr.rho < 1e30f) ||
                                     row_reach_alive(rr, ngx, ngy, ngz, make_float4(ea.x, ea.y, ea.z, rad[e]), es));
              }
              const uint64_t bal = __builtin_amdgcn_ballot_w64(al);
#ifdef PCLHIP_VERIFY_BOUNDS  // (row, entry) pairs the reach filter dropped: every lane of that row checks the entry's leaf
              {
                bool dropped = false;
                if (e < n_alive) dropped = !al && !(wl.list[LS * e + 1].w > T);
                for (uint64_t db = __builtin_amdgcn_ballot_w64(dropped); db != 0; db &= db - 1ull) {
                  const uint32_t bit = uint32_t(__builtin_ctzll(db));
                  const uint32_t vid = __float_as_uint(wl.list[LS * (e0 + (bit & 15u))].w);
                  PCLHIP_VERIFY_CULL(ix.soa, vid, qx[0], qy[0], qz[0], pol.worst(0),
                                     valid[0] && (uint32_t(lane) >> 4) == (bit >> 4) && pol_wants(pol, vid), ts);
                }
              }
#endif
              rowmask |= ((bal >> rshift) & 0xFFFFull) << e0;
            }
            // (lane, leaf) pairs: every lane walks the entries alive for its row
            uint64_t todo = valid[0] ? rowmask : 0ull, lanemask = 0;
            while (__builtin_amdgcn_ballot_w64(todo != 0) != 0) {
              ++ts.c[1];
              const bool has = todo != 0;
              const uint32_t e = has ? uint32_t(__builtin_ctzll(todo)) : b0;
              todo &= todo - 1ull;  // 0 stays 0
              const float4 ea = wl.list[LS * e], es = wl.list[LS * e + 2];
              const float lb = point_disc_lb(qx[0], qy[0], qz[0], make_float4(ea.x, ea.y, ea.z, rad[e]), es);
              const bool need = has && !(lb > pol.worst(0)) && pol_wants(pol, __float_as_uint(ea.w));
              PCLHIP_VERIFY_CULL(ix.soa, __float_as_uint(ea.w), qx[0], qy[0], qz[0], pol.worst(0),
                                 has && lb > pol.worst(0) && pol_wants(pol, __float_as_uint(ea.w)), ts);
              lanemask |= need ? (1ull << e) : 0ull;
            }
            // evaluation, 16 staged leaves at a time, every lane its own
            for (uint32_t c0 = b0; c0 < n_alive; c0 += LEAF_BATCH) {
              uint32_t m16 = uint32_t((lanemask >> c0) & 0xFFFFull);
              if (__builtin_amdgcn_ballot_w64(m16 != 0) == 0) continue;
              const uint32_t cn = (n_alive - c0) < uint32_t(LEAF_BATCH) ? (n_alive - c0) : uint32_t(LEAF_BATCH);
              stage(c0, cn);
              PCLHIP_WAIT_VMCNT0();
              while (__builtin_amdgcn_ballot_w64(m16 != 0) != 0) {
                uint32_t slot = 0, id = NO_INDEX;
                if (m16 != 0) {
                  slot = uint32_t(__builtin_ctz(m16));
                  id = __float_as_uint(wl.list[LS * (c0 + slot)].w);
                  m16 &= m16 - 1u;
                }
                round(slot, id);
              }
            }
            break;  // the list is done
          }
          stage(b0, nb);
          bool landed = false;
          if (loose) {
            // Loose bounds (first ICP iterations): walk the sorted batch; every lane keeps at most one
            // pending leaf and a round runs when some lane would need a second one, so bounds tighten
            // as early as possible and the tail of the list is cut off by the shrinking wave radius.
            uint32_t pslot = 0, pid = NO_INDEX;
            for (uint32_t t = 0; t < nb; ++t) {
              const float4 ea = wl.list[LS * (b0 + t)], eb = wl.list[LS * (b0 + t) + 1];  // broadcast reads
              if (uniform_f32(eb.w) > T) {  // sorted: every remaining leaf is farther than the wave radius
                cut = true;
                break;
              }
              ++ts.c[1];
              float lb;
              if (use_disc) {
                const float4 es = wl.list[LS * (b0 + t) + 2];
                lb = point_disc_lb(qx[0], qy[0], qz[0], make_float4(ea.x, ea.y, ea.z, rad[b0 + t]), es);
              } else {
                lb = point_box_lb(qx[0], qy[0], qz[0], ea.x, ea.y, ea.z, eb.x, eb.y, eb.z);
              }
              bool need = valid[0] && !(lb > pol.worst(0)) && pol_wants(pol, __float_as_uint(ea.w));
              PCLHIP_VERIFY_CULL(ix.soa, __float_as_uint(ea.w), qx[0], qy[0], qz[0], pol.worst(0),
                                 use_disc && valid[0] && lb > pol.worst(0) && pol_wants(pol, __float_as_uint(ea.w)), ts);
              if (__builtin_amdgcn_ballot_w64(need && pid != NO_INDEX) != 0) {
                if (!landed) {
                  PCLHIP_WAIT_VMCNT0();
                  landed = true;
                }
                const float w0 = pol.worst(0);
                round(pslot, pid);
                pid = NO_INDEX;
                const float now = pol.worst(0);
                if (__builtin_amdgcn_ballot_w64(valid[0] && now < w0) != 0) T = wave_max_f(valid[0] ? now : 0.0f);
                need = need && !(lb > now);
              }
              if (need) {
                pslot = t;
                pid = __float_as_uint(ea.w);
              }
            }
            if (__builtin_amdgcn_ballot_w64(pid != NO_INDEX) != 0) {
              if (!landed) {
                PCLHIP_WAIT_VMCNT0();
                landed = true;
              }
              const float w0 = pol.worst(0);
              round(pslot, pid);
              const float now = pol.worst(0);
              if (__builtin_amdgcn_ballot_w64(valid[0] && now < w0) != 0) T = wave_max_f(valid[0] ? now : 0.0f);
            }
          } else {
            // (Deferring these evaluations -- queueing the leaf ids per lane in LDS across batches and nodes and reading
            // the leaves straight from the index when a queue fills up or the search ends -- was measured: the rounds
            // then follow the largest number of leaves any one lane needs instead of the sum of the batches' maxima
            // (lanes are active in 14 of 64 slots of a round in the normals, 20 of 64 in a converged ICP iteration), but
            // every round waits for the index instead of finding its leaves staged under the box tests, and the bounds
            // tighten later: 0.565 -> 0.645 ms for a converged iteration, 2.07 -> 2.02 ms for the normals.)
            // Tight bounds (seeded iterations): one branch-free scan builds a per-lane bit mask of the
            // batch's leaves the lane's bound cannot exclude (the box tests pipeline freely), then every
            // lane pops its leaves in rounds.  Rounds per node ~ max over lanes of the leaves a lane
            // really needs (1-2), not the union over the wave's 64 queries.
            ts.c[1] += nb;
            uint32_t mask = 0;
            for (uint32_t t = 0; t < nb; ++t) {
              const float4 ea = wl.list[LS * (b0 + t)], eb = wl.list[LS * (b0 + t) + 1];  // broadcast reads
              const float lb = point_box_lb(qx[0], qy[0], qz[0], ea.x, ea.y, ea.z, eb.x, eb.y, eb.z);
              mask |= ((!(lb > pol.worst(0)) && pol_wants(pol, __float_as_uint(ea.w))) ? 1u : 0u) << t;
            }
            if (!valid[0]) mask = 0;
            while (__builtin_amdgcn_ballot_w64(mask != 0) != 0) {
              if (!landed) {
                PCLHIP_WAIT_VMCNT0();
                landed = true;
              }
              uint32_t slot = 0, id = NO_INDEX;
              if (mask != 0) {
                slot = uint32_t(__builtin_ctz(mask));
                id = __float_as_uint(wl.list[LS * (b0 + slot)].w);
                mask &= mask - 1u;
              }
              // (Re-testing a popped leaf against the lane's tightened bound, up to two pops per round, was measured: the
              // rounds of a batch did not get fewer -- 2.60 against 2.56 ms for the normals, +2 % on the seeded searches.)
              round(slot, id);
            }
          }
          // the next batch overwrites wl.buf: make sure this batch's DMA is not still landing
          if (!landed) PCLHIP_WAIT_VMCNT0();
        }
        {
          const float after = pol.worst(0);
          if (__builtin_amdgcn_ballot_w64(valid[0] && after < before) != 0) T = wave_max_f(valid[0] ? after : 0.0f);
        }
      } else {
      static_assert(SPARSE || LEAF_BATCH * LEAF_FLOATS <= WL::BUF_FLOATS, "wave-uniform evaluation stages whole leaves");
      bool cut = false;
      for (uint32_t b0 = 0; b0 < n_alive && !cut; b0 += LEAF_BATCH) {
        const uint32_t nb = (n_alive - b0) < uint32_t(LEAF_BATCH) ? (n_alive - b0) : uint32_t(LEAF_BATCH);
        // ---- stage the batch's candidate blocks: 16 x 16-byte chunks per leaf, LDS-DMA ----------
        // (all earlier reads of wl.buf were consumed by VALU work, so nothing is still in flight)
#pragma unroll
        for (int t = 0; t < (LEAF_BATCH * LEAF_FLOATS * 4) / (WAVE * 16); ++t) {  // 4 instructions
          const uint32_t slot = uint32_t(t) * (WAVE / 16) + uint32_t(lane) / 16u;
          if (slot < nb) {
            const uint32_t leaf_id = __float_as_uint(wl.list[LS * (b0 + slot)].w);
            const float* src = ix.soa + size_t(leaf_id) * LEAF_FLOATS + (lane & 15) * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(wl.buf + t * (WAVE * 4)), 16, 0,
                                             0);
          }
        }
        bool landed = false;
        for (uint32_t t = 0; t < nb; ++t) {
          const float4 ea = wl.list[LS * (b0 + t)], eb = wl.list[LS * (b0 + t) + 1];  // broadcast reads
          if (uniform_f32(eb.w) > T) {
            if (ordered) {  // sorted: every remaining leaf is farther than the wave radius
              cut = true;
              break;
            }
            continue;
          }
          ++ts.c[1];
          bool need = false;
#pragma unroll
          for (int q = 0; q < QPL; ++q) {
            float lb;
            if (use_disc) {
              const float4 es = wl.list[LS * (b0 + t) + 2];
              lb = point_disc_lb(qx[q], qy[q], qz[q], make_float4(ea.x, ea.y, ea.z, rad[b0 + t]), es);
            } else {
              lb = point_box_lb(qx[q], qy[q], qz[q], ea.x, ea.y, ea.z, eb.x, eb.y, eb.z);
            }
            need = need || (valid[q] && !(lb > pol.worst(q)));
            PCLHIP_VERIFY_CULL(ix.soa, __float_as_uint(ea.w), qx[q], qy[q], qz[q], pol.worst(q),
                               use_disc && valid[q] && lb > pol.worst(q), ts);
          }
          if (__builtin_amdgcn_ballot_w64(need) == 0) continue;
          ++ts.c[2];
          if (!landed) {  // first use of this batch: the DMA must have landed (it ran under the tests)
            PCLHIP_WAIT_VMCNT0();
            landed = true;
          }
          const float before = lane_worst(pol, valid);
          pol.leaf(wl.buf + t * LEAF_FLOATS, uniform_u32(__float_as_uint(ea.w)), qx, qy, qz);
          // the wave radius can only shrink if some lane's own bound shrank
          const float after = lane_worst(pol, valid);
          if (__builtin_amdgcn_ballot_w64(after < before) != 0) T = wave_max_f(after);
        }
        // the next batch overwrites wl.buf: make sure this batch's DMA is not still landing
        if (!landed) PCLHIP_WAIT_VMCNT0();
      }
      }  // wave-uniform leaf evaluation
    } else {
      int jn;
      if ((mask & (mask - 1)) == 0) {
        jn = __builtin_ctzll(mask);  // the usual case above the leaf level: exactly one survivor
      } else {
        const float m = wave_min_f(alive ? lbG : INF);
        jn = __builtin_ctzll(__builtin_amdgcn_ballot_w64(alive && lbG == m));
      }
      const uint64_t others = mask & ~(1ull << jn);
      if (ordered && __builtin_popcountll(others) > 1) {
        // push the other survivors farthest first so that pops come back nearest first
        // same unique keys (see the leaf level); the complement of the lane makes equal bounds pop in lane order
        const uint32_t key = (__float_as_uint(lbG) & ~63u) | uint32_t(63 - lane);
        uint32_t rank = 0;
        for (uint64_t m2 = others; m2; m2 &= m2 - 1) {
          const uint32_t sk = uint32_t(__builtin_amdgcn_readlane(int(key), __builtin_ctzll(m2)));
          rank += sk > key ? 1u : 0u;
        }
        if (alive && lane != jn)
          stack[sp + int(rank)] = make_uint2((cl << 28) | (first + uint32_t(lane)), __float_as_uint(lbG));
      } else if (alive && lane != jn) {
        const int at = sp + __builtin_popcountll(others & ((1ull << lane) - 1ull));
        stack[at] = make_uint2((cl << 28) | (first + uint32_t(lane)), __float_as_uint(lbG));
      }
      sp += __builtin_popcountll(others);
#ifndef PCLHIP_STATS_LANES
      ts.c[3] += uint32_t(__builtin_popcountll(others));
#endif
      __builtin_amdgcn_wave_barrier();
      level = cl;
      node = first + uint32_t(jn);
      have = true;
    }
  }
}

__device__ __forceinline__ void flush_stats(const TraverseStats& ts_in, unsigned long long* g) {
#ifdef PCLHIP_VERIFY_BOUNDS  // slots 6 / 7 were counted per lane (verify_culled_leaf)
  TraverseStats ts = ts_in;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ts.c[6] += __shfl_xor(ts.c[6], o);
    ts.c[7] += __shfl_xor(ts.c[7], o);
  }
#else
  const TraverseStats& ts = ts_in;
#endif
  if (g != nullptr && (threadIdx.x & (WAVE - 1)) == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(g + i, (unsigned long long)ts.c[i]);
  }
}

// XCD-aware mapping of (block, wave, iteration) -> query group: blocks that share an XCD (and its L2) work on one
// contiguous window of kd-ordered groups at a time; the waves of an XCD interleave over the window.
struct GroupSchedule {
  uint32_t groups_per_xcd, xcd_first, slot_wave, waves_per_xcd;
  __device__ __forceinline__ GroupSchedule(uint32_t ngroups) {
    const uint32_t nxcd = gridDim.x < 8u ? gridDim.x : 8u;  // partitions = XCDs that own a block
    const uint32_t xcd = blockIdx.x % nxcd, slot = blockIdx.x / nxcd;
    const uint32_t slots = (gridDim.x + nxcd - 1 - xcd) / nxcd;  // blocks on this XCD
    const uint32_t waves_per_block = blockDim.x / WAVE;
    groups_per_xcd = (ngroups + nxcd - 1) / nxcd;
    xcd_first = xcd * groups_per_xcd;
    slot_wave = slot * waves_per_block + threadIdx.x / WAVE;
    waves_per_xcd = slots * waves_per_block;
  }
  __device__ __forceinline__ uint32_t first() const { return slot_wave; }
  __device__ __forceinline__ uint32_t step() const { return waves_per_xcd; }
  __device__ __forceinline__ uint32_t end() const { return groups_per_xcd; }  // local groups [first, end) in steps of step()
  // local group index g (0 <= g < groups_per_xcd) -> global group
  __device__ __forceinline__ uint32_t global(uint32_t g) const { return xcd_first + g; }
};

// The same window, balanced.  With fixed shares a launch ends when its slowest wave does -- 38 groups per wave at 10M
// points, each between half and twice the mean: the normals kernel ran 1.86 ms where its waves were busy for 1.5.  A wave
// strides over the first HALF of its XCD's groups like GroupSchedule (equal shares, no traffic), then takes the remaining
// ones one at a time from the XCD's counter: whoever is through first takes more, and the waves of an XCD still work on
// adjacent groups.  The counters (IndexView::sched_ctr; zeroed in stream order by PCLHIP_LAUNCH_FED) are asked one group
// AHEAD of use -- ahead() at the top of a group, advance() at its bottom -- so the atomic's round trip is off the
// critical path.  Measured at 10M points (static share 7/8, 3/4, 1/2, one round): normals 1.76 / 1.67 / 1.49 / 1.50 ms,
// ms per ICP step 1.28 / 1.244 / 1.243 / 1.26 (1.326 with fixed shares).
struct GroupFeed {
  static constexpr uint32_t END = 0xFFFFFFFFu;
  uint32_t step_, static_end_, end_;
  uint32_t* ctr_;
  uint32_t ticket_, next_;
  bool asked_;
  __device__ __forceinline__ GroupFeed(const GroupSchedule& s, uint32_t* counters) {
    step_ = s.step();
    end_ = s.end();
#ifndef PCLHIP_DYN_STATIC_8THS
#define PCLHIP_DYN_STATIC_8THS 4
#endif
    const uint32_t all = end_ / step_;  // whole strides of the window
    uint32_t rounds = all * uint32_t(PCLHIP_DYN_STATIC_8THS) / 8u;  // ... of the static part: at least one (a wave's first group)
    if (rounds < 1u) rounds = 1u;
    static_end_ = (counters != nullptr && all >= 2u) ? rounds * step_ : end_;
    const uint32_t nxcd = gridDim.x < 8u ? gridDim.x : 8u;
    ctr_ = counters + (blockIdx.x % nxcd) * uint32_t(SCHED_CTR_STRIDE);
    ticket_ = 0;
    next_ = END;
    asked_ = false;
  }
  // the wave's first group (local index), END if it has none
  __device__ __forceinline__ uint32_t first(const GroupSchedule& s) const { return s.first() < end_ ? s.first() : END; }
  // groups per wave of the launch (whole strides of the XCD's window)
  __device__ __forceinline__ uint32_t strides() const { return end_ / step_; }
  // the group after `gl` when it is known without asking: true and `out` (END: none); false: request() + resolve()
  __device__ __forceinline__ bool static_next(uint32_t gl, uint32_t& out) const {
    if (static_end_ == end_) {
      out = (gl + step_ < end_) ? gl + step_ : END;
      return true;
    }
    if (gl < static_end_ && gl + step_ < static_end_) {
      out = gl + step_;
      return true;
    }
    return false;
  }
  __device__ __forceinline__ void request() {
    if ((threadIdx.x & (WAVE - 1)) == 0) ticket_ = atomicAdd(ctr_, 1u);
  }
  __device__ __forceinline__ uint32_t resolve() const {
    const uint32_t g = static_end_ + uniform_u32(ticket_);
    return g < end_ ? g : END;
  }
  // simple loops: for (gl = feed.first(sched); gl != END; gl = feed.advance()) { ...; feed.ahead(gl); ... }
  __device__ __forceinline__ void ahead(uint32_t gl) {
    asked_ = !static_next(gl, next_);
    if (asked_) request();
  }
  __device__ __forceinline__ uint32_t advance() const { return asked_ ? resolve() : next_; }
};

}  // namespace pclhip
