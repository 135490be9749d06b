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

#include "traverse.hpp"

namespace pclhip {

constexpr uint32_t SO_LIST_CAP = 512;   // collected leaf ids (9 bits of the row-seed keys)
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

// The previous group of this wave: its queries (as searched) and their matches -- where the seed comes from.
struct PrevGroup {
  float x, y, z;
  uint32_t pos;  // sorted position of the lane's match, NO_INDEX if none
};

// Returns true when the group is done (pol holds every valid lane's exact minimum, up to the tie / resolve steps the
// caller runs anyway); false when the caller has to run traverse() -- pol then holds valid bounds and seeds.
// `seed_leaf_out`: leaf of the seed (a start hint for the fallback).  Must be called by all 64 lanes.
template <class WL>
__device__ __forceinline__ bool standoff_search(const IndexView& ix, float qx, float qy, float qz, bool valid, NN1Min& pol,
                                                WL& wl, const Box* topbox, TraverseStats& ts, const PrevGroup& prev,
                                                bool allow_skip, uint32_t& seed_leaf_out) {
  static_assert(WL::BUF_FLOATS >= 768, "staging / id area: 3 KB");
  const int lane = threadIdx.x & (WAVE - 1);
  const uint32_t sub = uint32_t(lane) & 15u, row = uint32_t(lane) >> 4;
  const float BIG = 3.402823466e+38f;
  const float INF = __builtin_inff();
  seed_leaf_out = NO_INDEX;
  if (__builtin_amdgcn_ballot_w64(valid) == 0 || ix.n == 0) return true;
  const float qxa[1] = {qx}, qya[1] = {qy}, qza[1] = {qz};

  // ---- group box -------------------------------------------------------------------------------------------------
  float lx0 = valid ? qx : BIG, ly0 = valid ? qy : BIG, lz0 = valid ? qz : BIG;
  float hx0 = valid ? qx : -BIG, hy0 = valid ? qy : -BIG, hz0 = valid ? qz : -BIG, dummy = 0.0f;
  wave_min3_max4(lx0, ly0, lz0, hx0, hy0, hz0, dummy);
  const float Qlx = lx0, Qly = ly0, Qlz = lz0, Qhx = hx0, Qhy = hy0, Qhz = hz0;

  // ---- 0. seed ---------------------------------------------------------------------------------------------------
  uint32_t seed_pos = NO_INDEX;
  {
    const float Cx = 0.5f * (Qlx + Qhx), Cy = 0.5f * (Qly + Qhy), Cz = 0.5f * (Qlz + Qhz);
    const bool pok = prev.pos != NO_INDEX;
    if (__builtin_amdgcn_ballot_w64(pok) != 0) {
      const float dx = prev.x - Cx, dy = prev.y - Cy, dz = prev.z - Cz;
      const float d = pok ? (dx * dx + dy * dy) + dz * dz : INF;
      const float m = wave_min_f(d);
      const uint64_t at = __builtin_amdgcn_ballot_w64(pok && d == m);
      if (at != 0) seed_pos = uint32_t(__builtin_amdgcn_readlane(int(prev.pos), __builtin_ctzll(at)));
    }
    if (seed_pos == NO_INDEX) {
      // no previous group (first of the wave's chunk) or it had no match: the exact neighbour of ONE lane
      const uint64_t vm = __builtin_amdgcn_ballot_w64(valid);
      const bool one[1] = {valid && lane == __builtin_ctzll(vm)};
      traverse<NN1Min, true>(ix, qxa, qya, qza, one, pol, wl, topbox, ts, NO_INDEX, true);
      pol.resolve(ix, qxa, qya, qza);
      seed_pos = uint32_t(__builtin_amdgcn_readlane(int(pol.bestpos[0]), __builtin_ctzll(vm)));
    }
  }
  if (seed_pos != NO_INDEX) {
    seed_leaf_out = seed_pos / LEAF;
    const float4 t = ix.pts[seed_pos];  // wave-uniform address
    if (valid) pol.seed(0, l2_simple(qx, qy, qz, t.x, t.y, t.z), seed_pos);
  }
  const float T = wave_max_f(valid ? pol.worst(0) : 0.0f);  // wave radius (squared); fixed from here on
  if (!(T < INF)) return false;
  if (!(T > ix.disc_from)) return false;  // next to the surface a disc excludes nothing a box does not: traverse()

  uint2* const stack = wl.stack;
  uint32_t* const ids = reinterpret_cast<uint32_t*>(wl.buf);
  float* const area_b = reinterpret_cast<float*>(wl.list);  // 3 KB: staging of the row seeds, then the union slots

  // ---- 1. collect ------------------------------------------------------------------------------------------------
  uint32_t level = uint32_t(ix.top) + 1u, node = 0u;  // virtual root above the top level
  if (allow_skip && seed_leaf_out != NO_INDEX) {       // start below the root when the search ball fits (see traverse())
    const auto inside = [&](const Box& b) {
      const float d = fminf(fminf(fminf(Qlx - b.lo.x, b.hi.x - Qhx), fminf(Qly - b.lo.y, b.hi.y - Qhy)),
                            fminf(Qlz - b.lo.z, b.hi.z - Qhz));
      return d > 0.0f && d * d * 0.999999f > T;
    };
    const bool has2 = ix.top >= 2, has3 = ix.top >= 3;
    Box b2 = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)}, b3 = b2;
    if (has2) b2 = ix.box[2][seed_leaf_out >> 6];
    if (has3) b3 = ix.box[3][seed_leaf_out >> 12];
    if (has2 && inside(b2)) {
      level = 2u;
      node = seed_leaf_out >> 6;
    } else if (has3 && inside(b3)) {
      level = 3u;
      node = seed_leaf_out >> 12;
    }
  }
  uint32_t n = 0;
  {
    int sp = 0;
    bool have = true;
    for (;;) {
      if (!have) {
        if (sp == 0) break;
        --sp;
        __builtin_amdgcn_wave_barrier();
        const uint32_t ex = uniform_u32(stack[sp].x);
        level = ex >> 28;
        node = ex & 0x0FFFFFFFu;
      }
      have = false;
      ++ts.c[0];
      const uint32_t cl = level - 1u;  // level of the children
      const uint32_t first = node * FANOUT;
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
      const float lbG = has ? box_box_lb(Qlx, Qly, Qlz, Qhx, Qhy, Qhz, lx, ly, lz, hx, hy, hz) : INF;
      const bool alive = has && !(lbG > T);
      const uint64_t mask = __builtin_amdgcn_ballot_w64(alive);
      if (mask == 0) continue;
      const uint32_t cnt = uint32_t(__builtin_popcountll(mask));
      const uint32_t pre = uint32_t(__builtin_popcountll(mask & ((1ull << lane) - 1ull)));
      if (cl == 1u) {
        if (n + cnt > SO_LIST_CAP) return false;
        if (alive) ids[n + pre] = first + uint32_t(lane);
        n += cnt;
      } else {
        if (alive) stack[sp + int(pre)] = make_uint2((cl << 28) | (first + uint32_t(lane)), 0u);
        sp += int(cnt);
        ts.c[3] += cnt;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (n == 0) {  // nothing within the wave radius (a finite maximum distance and no seed inside it)
    ++ts.c[4];
    return true;
  }

  // ---- 2. row seeds ----------------------------------------------------------------------------------------------
  const RowGeom geo = row_geometry(qx, qy, qz, valid);
  uint32_t id1 = NO_INDEX, id2 = NO_INDEX;
  {
    uint32_t m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu;  // the lane's two smallest (distance | entry) keys
    for (uint32_t e0 = 0; e0 < n; e0 += 16u) {
      const uint32_t e = e0 + sub;
      if (e < n) {
        const float4 c = ix.disc[2 * size_t(ids[e])];
        const float dx = geo.cx - c.x, dy = geo.cy - c.y, dz = geo.cz - c.z;
        const float d = (dx * dx + dy * dy) + dz * dz;  // >= 0: its bit pattern orders like the value
        const uint32_t key = (__float_as_uint(d) & ~0x1FFu) | e;
        m2 = min(m2, max(m1, key));
        m1 = min(m1, key);
      }
    }
    const uint32_t r1 = row_min_u32(m1);
    const uint32_t r2 = row_min_u32(m1 == r1 ? m2 : m1);
    if (geo.any && r1 != 0xFFFFFFFFu) id1 = ids[r1 & 0x1FFu];
    if (geo.any && r2 != 0xFFFFFFFFu) id2 = ids[r2 & 0x1FFu];
    // slot 2 * row + k <- the row's k-th seed leaf; the ids travel through the (now idle) stack area
    uint32_t* const tmp = reinterpret_cast<uint32_t*>(stack);
    if (sub == 0u) {
      tmp[2 * row] = id1;
      tmp[2 * row + 1] = id2;
    }
    __builtin_amdgcn_wave_barrier();
    so_stage(ix, area_b, sub < 8u ? tmp[sub] : NO_INDEX);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (uint32_t k = 0; k < 2; ++k) {
      const uint32_t id = valid ? (k == 0 ? id1 : id2) : NO_INDEX;
      if (__builtin_amdgcn_ballot_w64(id != NO_INDEX) != 0) {
        ++ts.c[2];
        pol.leaf_lane(area_b, 2 * row + k, id, qxa, qya, qza);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }

  // ---- 3. row cull: (row, leaf) pairs, 64 per pass; leaves alive for any row -> union slots ------------------------
  const RowReach rr = row_reach_of(geo, qx, qy, qz, valid, pol.worst(0));
  uint32_t ucnt = 0;
  uint64_t rowmask;
  {
    uint32_t acc_lo = 0, acc_hi = 0;  // this lane's own alive pairs as union-slot bits
    for (uint32_t e0 = 0; e0 < n; e0 += 16u) {
      const uint32_t e = e0 + sub;
      bool al = false;
      uint32_t id = 0;
      float4 cR = make_float4(0, 0, 0, 0), nh = cR;
      if (e < n) {
        id = ids[e];
        cR = ix.disc[2 * size_t(id)];
        nh = ix.disc[2 * size_t(id) + 1];
        al = geo.any && (!(rr.rho < 1e30f) || row_reach_alive(rr, geo.ngx, geo.ngy, geo.ngz, cR, nh));
      }
      const uint64_t bal = __builtin_amdgcn_ballot_w64(al);
      if (bal == 0) continue;
      const uint32_t any16 = uint32_t((bal | (bal >> 16) | (bal >> 32) | (bal >> 48)) & 0xFFFFull);
      const uint32_t cnt = uint32_t(__builtin_popcount(any16));
      if (ucnt + cnt > SO_UNION_CAP) return false;
      const uint32_t u = ucnt + uint32_t(__builtin_popcount(any16 & ((1u << sub) - 1u)));
      // the same entry is held by one lane of every row: the lowest alive row writes the slot
      const uint64_t below = bal & ((1ull << lane) - 1ull) & (0x0001000100010001ull << sub);
      if (al && below == 0) {
        wl.list[3 * u] = make_float4(cR.x, cR.y, cR.z, __uint_as_float(id));
        wl.list[3 * u + 2] = nh;
        wl.rad[u] = cR.w;
      }
      if (al) {  // u < 64
        const uint64_t bit = 1ull << u;
        acc_lo |= uint32_t(bit);
        acc_hi |= uint32_t(bit >> 32);
      }
      ucnt += cnt;
    }
    rowmask = (uint64_t(row_or_u32(acc_hi)) << 32) | uint64_t(row_or_u32(acc_lo));
  }
  __builtin_amdgcn_wave_barrier();

  // ---- 4. lane cull: every lane against the slots alive for its row --------------------------------------------------
  uint64_t lanemask = 0;
  {
    uint64_t todo = valid ? rowmask : 0ull;
    while (__builtin_amdgcn_ballot_w64(todo != 0) != 0) {
      ++ts.c[1];
      const bool has = todo != 0;
      const uint32_t e = has ? uint32_t(__builtin_ctzll(todo)) : 0u;
      todo &= todo - 1ull;  // 0 stays 0
      const float4 ea = wl.list[3 * e], es = wl.list[3 * e + 2];
      const float lb = point_disc_lb(qx, qy, qz, make_float4(ea.x, ea.y, ea.z, wl.rad[e]), es);
      const uint32_t id = __float_as_uint(ea.w);
      const bool need = has && !(lb > pol.worst(0)) && id != id1 && id != id2;  // the row seeds are done
      lanemask |= need ? (1ull << e) : 0ull;
    }
  }

  // ---- 5. evaluation, 16 staged leaves at a time, every lane its own -------------------------------------------------
  for (uint32_t c0 = 0; c0 < ucnt; c0 += LEAF_BATCH) {
    uint32_t m16 = uint32_t((lanemask >> c0) & 0xFFFFull);
    if (__builtin_amdgcn_ballot_w64(m16 != 0) == 0) continue;
    const uint32_t cn = (ucnt - c0) < uint32_t(LEAF_BATCH) ? (ucnt - c0) : uint32_t(LEAF_BATCH);
    so_stage(ix, wl.buf, sub < cn ? __float_as_uint(wl.list[3 * (c0 + sub)].w) : NO_INDEX);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    while (__builtin_amdgcn_ballot_w64(m16 != 0) != 0) {
      uint32_t slot = 0, id = NO_INDEX;
      if (m16 != 0) {
        slot = uint32_t(__builtin_ctz(m16));
        id = __float_as_uint(wl.list[3 * (c0 + slot)].w);
        m16 &= m16 - 1u;
      }
      ++ts.c[2];
      pol.leaf_lane(wl.buf, slot, id, qxa, qya, qza);
    }
  }
  ++ts.c[4];
  return true;
}

}  // namespace pclhip
