// traverse.hpp -- wavefront-cooperative exact nearest-neighbour traversal of the implicit wide BVH.
//
// One wavefront (64 lanes) owns 64 spatially coherent queries (consecutive in Morton order), one
// per lane.  Tree nodes are visited by the WAVE, not by lanes:
//   * interior node: lane j loads child box j (one coalesced 2 KB read), tests it against the
//     bounding box of the wave's 64 queries and the wave's pruning radius T = max_i worst_i;
//     survivors are pushed on a wave-uniform stack in LDS, the nearest child is entered first.
//   * leaf (16 consecutive sorted points): every lane tests its own query against the leaf box;
//     if any lane still needs the leaf, its 16 candidates are read through wave-uniform (scalar)
//     loads, i.e. they sit in SGPRs and every lane evaluates all 16 distances -- coalesced /
//     broadcast loads only, no per-lane gathers, no divergence inside the hot loop.
// All bounds are exact in float: rounding is monotone and the bound uses the same operation order
// as the distance, so box_lb(q, box) <= l2_simple(q, c) for every c in the box, bit for bit.
// Distances follow FLANN's L2_Simple order ((dx*dx)+dy*dy)+dz*dz with no FMA contraction
// (call sites kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp:154-203).
#pragma once

#include "pclhip_internal.hpp"

namespace pclhip {

constexpr int STACK_ENTRIES = 64 * MAX_LEVELS;  // worst case: every level pushes 63 siblings

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

// ---- wavefront reductions (all 64 lanes must be active) ---------------------------------------
// DPP butterflies (no LDS traffic, plain VALU latency): quad_perm swaps, row_half_mirror and
// row_mirror leave every row of 16 lanes holding its row result; row_bcast15/31 then fold the four
// rows into lane 63.  The result is read back with readlane so the compiler knows it is
// wave-uniform (SGPR): every branch on it is a scalar branch.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_min_f(float v) {
  v = fminf(v, dpp_f<0xB1, 0xf>(v));   // quad_perm [1,0,3,2]
  v = fminf(v, dpp_f<0x4E, 0xf>(v));   // quad_perm [2,3,0,1]
  v = fminf(v, dpp_f<0x141, 0xf>(v));  // row_half_mirror
  v = fminf(v, dpp_f<0x140, 0xf>(v));  // row_mirror
  v = fminf(v, dpp_f<0x142, 0xa>(v));  // row_bcast15 -> rows 1,3
  v = fminf(v, dpp_f<0x143, 0xc>(v));  // row_bcast31 -> rows 2,3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_f(float v) {
  v = fmaxf(v, dpp_f<0xB1, 0xf>(v));
  v = fmaxf(v, dpp_f<0x4E, 0xf>(v));
  v = fmaxf(v, dpp_f<0x141, 0xf>(v));
  v = fmaxf(v, dpp_f<0x140, 0xf>(v));
  v = fmaxf(v, dpp_f<0x142, 0xa>(v));
  v = fmaxf(v, dpp_f<0x143, 0xc>(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
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

// Leaf candidates are read through the CONSTANT address space: the address is wave-uniform and the
// index is never written while a search kernel runs, so hipcc emits s_load_dwordx4 (scalar cache
// -> SGPRs) and every lane gets the candidate as a scalar operand -- no VGPRs, no vector-memory
// instructions and no LDS traffic in the all-pairs loop.
typedef const float __attribute__((address_space(4))) * leaf_ptr_t;  // 4 floats per candidate
__device__ __forceinline__ leaf_ptr_t leaf_pointer(const float4* pts, uint32_t leaf_id) {
  return (leaf_ptr_t)(unsigned long long)(pts + size_t(leaf_id) * LEAF);
}

constexpr uint64_t KEY_NONE = (uint64_t(0x7F800000u) << 32) | 0xFFFFFFFFull;  // (+inf, no index)
__device__ __forceinline__ uint64_t make_key(float d, uint32_t idx) {
  return (uint64_t(__float_as_uint(d)) << 32) | idx;
}
__device__ __forceinline__ float key_dist(uint64_t k) { return __uint_as_float(uint32_t(k >> 32)); }
__device__ __forceinline__ uint32_t key_index(uint64_t k) { return uint32_t(k); }

// ---- leaf policies -------------------------------------------------------------------------------
// 1-NN: best (distance, original index) key + sorted position of the winner.
struct NN1 {
  uint64_t key;
  uint32_t pos;
  __device__ __forceinline__ void init(uint64_t k0) {
    key = k0;
    pos = NO_INDEX;
  }
  __device__ __forceinline__ float worst() const { return key_dist(key); }
  __device__ __forceinline__ void leaf(const IndexView& ix, uint32_t leaf_id, float qx, float qy, float qz) {
    const leaf_ptr_t p = leaf_pointer(ix.pts, leaf_id);  // wave-uniform address
    const uint32_t base = leaf_id * LEAF;
#pragma unroll
    for (int c = 0; c < LEAF; ++c) {
      const float d = l2_simple(qx, qy, qz, p[4 * c], p[4 * c + 1], p[4 * c + 2]);
      const uint64_t k = make_key(d, __float_as_uint(p[4 * c + 3]));
      const bool t = k < key;
      key = t ? k : key;
      pos = t ? base + c : pos;
    }
  }
};

// 1-NN fast path: the hot loop only tracks the minimum DISTANCE (packed v_pk_* math on candidate
// pairs read as SGPR pairs from the per-leaf SoA copy, one v_min3 per pair) and remembers in which
// leaf the minimum was first reached.  The winner's index is resolved once per query afterwards
// (resolve()); exact cross-leaf distance ties raise `tie`, and the caller then re-runs the exact
// (distance, index) policy NN1 for those lanes, so results stay bit-identical to the oracle.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const v2f __attribute__((address_space(4))) * soa_ptr_t;
struct NN1Min {
  float best;         // candidates must be strictly below this to win
  uint32_t bestleaf;  // leaf that first reached `best`, NO_INDEX while best is only the bound
  bool tie;
  __device__ __forceinline__ void init(float bound_exclusive) {
    best = bound_exclusive;
    bestleaf = NO_INDEX;
    tie = false;
  }
  __device__ __forceinline__ void seed(float d, uint32_t leaf_id) {
    if (d < best) {
      best = d;
      bestleaf = leaf_id;
    }
  }
  __device__ __forceinline__ float worst() const { return best; }
  __device__ __forceinline__ void leaf(const IndexView& ix, uint32_t leaf_id, float qx, float qy, float qz) {
    const soa_ptr_t p = (soa_ptr_t)(unsigned long long)(ix.soa + size_t(leaf_id) * (3 * LEAF));
    const v2f qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
    float m = __builtin_inff();
#pragma unroll
    for (int j = 0; j < LEAF / 2; ++j) {
      const v2f dx = qx2 - p[j], dy = qy2 - p[LEAF / 2 + j], dz = qz2 - p[LEAF + j];
      v2f r = dx * dx;
      r = r + dy * dy;
      r = r + dz * dz;
      m = __builtin_fminf(m, __builtin_fminf(r.x, r.y));
    }
    const bool imp = m < best;
    const bool eq = (m == best) && (bestleaf != NO_INDEX) && (bestleaf != leaf_id);
    tie = imp ? false : (tie || eq);
    bestleaf = imp ? leaf_id : bestleaf;
    best = imp ? m : best;
  }
  // (distance, original index) key and sorted position of the winner; NO_INDEX if none
  __device__ __forceinline__ void resolve(const IndexView& ix, float qx, float qy, float qz, uint64_t& key,
                                          uint32_t& pos) const {
    key = KEY_NONE;
    pos = NO_INDEX;
    if (bestleaf == NO_INDEX) return;
    const float4* __restrict__ p = ix.pts + size_t(bestleaf) * LEAF;
#pragma unroll 4
    for (int c = 0; c < LEAF; ++c) {
      const float4 v = p[c];
      const float d = l2_simple(qx, qy, qz, v.x, v.y, v.z);
      const uint64_t k = make_key(d, __float_as_uint(v.w));
      if (d == best && k < key) {
        key = k;
        pos = bestleaf * LEAF + c;
      }
    }
  }
};

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
  __device__ __forceinline__ float worst() const { return key_dist(keys[K - 1]); }
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
  __device__ __forceinline__ void leaf(const IndexView& ix, uint32_t leaf_id, float qx, float qy, float qz) {
    const leaf_ptr_t p = leaf_pointer(ix.pts, leaf_id);
    const uint32_t base = leaf_id * LEAF;
#pragma unroll
    for (int c = 0; c < LEAF; ++c) {
      const float d = l2_simple(qx, qy, qz, p[4 * c], p[4 * c + 1], p[4 * c + 2]);
      const uint64_t k = make_key(d, __float_as_uint(p[4 * c + 3]));
      if (__builtin_amdgcn_ballot_w64(k < keys[K - 1]) != 0) insert(k, base + c);  // wave-uniform branch
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
  __device__ __forceinline__ float worst() const { return key_dist(root); }
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
  __device__ __forceinline__ void leaf(const IndexView& ix, uint32_t leaf_id, float qx, float qy, float qz) {
    const leaf_ptr_t p = leaf_pointer(ix.pts, leaf_id);
    for (int c = 0; c < LEAF; ++c) {
      const float d = l2_simple(qx, qy, qz, p[4 * c], p[4 * c + 1], p[4 * c + 2]);
      const uint64_t key = make_key(d, __float_as_uint(p[4 * c + 3]));
      if (key < root) replace_root(key);
    }
  }
  // in-place heap sort -> ascending keys in heap[0..k)
  __device__ void sort_ascending() {
    for (int end = k - 1; end > 0; --end) {
      const uint64_t top = heap[0];
      const uint64_t last = heap[size_t(end) * stride];
      heap[size_t(end) * stride] = top;
      // sift `last` down in heap[0..end)
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
// that passed the per-lane test (= 16-candidate all-pairs blocks), [3] stack pushes, [4] groups.
struct TraverseStats {
  uint32_t c[5] = {0, 0, 0, 0, 0};
};

// ---- the traversal --------------------------------------------------------------------------------
// `stack` points at this wave's STACK_ENTRIES uint2 slots in LDS.  Must be called by all 64 lanes.
template <class Policy>
__device__ __forceinline__ void traverse(const IndexView& ix, float qx, float qy, float qz, bool valid,
                                         Policy& pol, uint2* stack, TraverseStats& ts) {
  const int lane = threadIdx.x & (WAVE - 1);
  if (__builtin_amdgcn_ballot_w64(valid) == 0 || ix.n == 0) return;
  ++ts.c[4];
  const float BIG = 3.402823466e+38f;
  const float INF = __builtin_inff();
  // bounding box of the wave's queries
  const float Qlx = wave_min_f(valid ? qx : BIG), Qly = wave_min_f(valid ? qy : BIG),
              Qlz = wave_min_f(valid ? qz : BIG);
  const float Qhx = wave_max_f(valid ? qx : -BIG), Qhy = wave_max_f(valid ? qy : -BIG),
              Qhz = wave_max_f(valid ? qz : -BIG);
  float T = wave_max_f(valid ? pol.worst() : 0.0f);  // wave pruning radius (squared)
  const float gdiag2 = (Qhx - Qlx) * (Qhx - Qlx) + (Qhy - Qly) * (Qhy - Qly) + (Qhz - Qlz) * (Qhz - Qlz);

  int sp = 0;
  uint32_t level = uint32_t(ix.top) + 1u, node = 0u;  // virtual root above the top level
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
    const uint32_t total = ix.count[cl];
    const uint32_t nchild = (total - first) < uint32_t(FANOUT) ? (total - first) : uint32_t(FANOUT);
    const bool has = uint32_t(lane) < nchild;
    float lx = 0, ly = 0, lz = 0, hx = 0, hy = 0, hz = 0;
    if (has) {
      const Box b = ix.box[cl][first + lane];
      lx = b.lo.x; ly = b.lo.y; lz = b.lo.z;
      hx = b.hi.x; hy = b.hi.y; hz = b.hi.z;
    }
    const float lbG = has ? box_box_lb(Qlx, Qly, Qlz, Qhx, Qhy, Qhz, lx, ly, lz, hx, hy, hz) : INF;
    const bool alive = has && !(lbG > T);
    uint64_t mask = __builtin_amdgcn_ballot_w64(alive);
    if (mask == 0) continue;
    // Visiting order.  Cold or lukewarm bounds (wave radius not yet small against the group's own
    // extent, e.g. the first ICP iterations): visit children in ascending order of their distance
    // to the group box, so the bounds collapse after the first few leaves and everything farther
    // is cut off at once.  Tight bounds (seeded steady state): the nearest child first, then plain
    // index order -- no per-child reduction.
    const bool ordered = T * 16.0f > gdiag2;
    if (cl == 1u) {
      if (ordered) {
        bool live = alive;
        for (;;) {
          if (__builtin_amdgcn_ballot_w64(live) == 0) break;
          const float m = wave_min_f(live ? lbG : INF);
          if (m > T) break;  // every remaining leaf is farther than the wave radius
          const int j = __builtin_ctzll(__builtin_amdgcn_ballot_w64(live && lbG == m));
          if (lane == j) live = false;
          ++ts.c[1];
          const float blx = readlane_f(lx, j), bly = readlane_f(ly, j), blz = readlane_f(lz, j);
          const float bhx = readlane_f(hx, j), bhy = readlane_f(hy, j), bhz = readlane_f(hz, j);
          const float lb = point_box_lb(qx, qy, qz, blx, bly, blz, bhx, bhy, bhz);
          const bool need = valid && !(lb > pol.worst());
          if (__builtin_amdgcn_ballot_w64(need) == 0) continue;
          ++ts.c[2];
          const float before = pol.worst();
          pol.leaf(ix, uniform_u32(first + uint32_t(j)), qx, qy, qz);
          if (__builtin_amdgcn_ballot_w64(valid && pol.worst() < before) != 0)
            T = wave_max_f(valid ? pol.worst() : 0.0f);
        }
      } else {
        const float m = wave_min_f(alive ? lbG : INF);
        const int jn = __builtin_ctzll(__builtin_amdgcn_ballot_w64(alive && lbG == m));
        bool first_done = false;
        while (mask) {
          int j;
          if (!first_done) {
            j = jn;
            first_done = true;
          } else {
            j = __builtin_ctzll(mask);
          }
          mask &= ~(1ull << j);
          if (readlane_f(lbG, j) > T) continue;
          ++ts.c[1];
          const float blx = readlane_f(lx, j), bly = readlane_f(ly, j), blz = readlane_f(lz, j);
          const float bhx = readlane_f(hx, j), bhy = readlane_f(hy, j), bhz = readlane_f(hz, j);
          const float lb = point_box_lb(qx, qy, qz, blx, bly, blz, bhx, bhy, bhz);
          const bool need = valid && !(lb > pol.worst());
          if (__builtin_amdgcn_ballot_w64(need) == 0) continue;
          ++ts.c[2];
          const float before = pol.worst();
          pol.leaf(ix, uniform_u32(first + uint32_t(j)), qx, qy, qz);
          // the wave radius can only shrink if some lane's own bound shrank
          if (__builtin_amdgcn_ballot_w64(valid && pol.worst() < before) != 0)
            T = wave_max_f(valid ? pol.worst() : 0.0f);
        }
      }
    } else {
      const float m = wave_min_f(alive ? lbG : INF);
      const int jn = __builtin_ctzll(__builtin_amdgcn_ballot_w64(alive && lbG == m));
      const uint64_t others = mask & ~(1ull << jn);
      if (ordered && __builtin_popcountll(others) > 1) {
        // push the other survivors farthest first so that pops come back nearest first
        bool live = alive && lane != jn;
        int at = sp;
        for (;;) {
          if (__builtin_amdgcn_ballot_w64(live) == 0) break;
          const float far = wave_max_f(live ? lbG : -1.0f);
          const int j = __builtin_ctzll(__builtin_amdgcn_ballot_w64(live && lbG == far));
          if (lane == j) {
            live = false;
            stack[at] = make_uint2((cl << 28) | (first + uint32_t(lane)), __float_as_uint(lbG));
          }
          ++at;
        }
      } else if (alive && lane != jn) {
        const int at = sp + __builtin_popcountll(others & ((1ull << lane) - 1ull));
        stack[at] = make_uint2((cl << 28) | (first + uint32_t(lane)), __float_as_uint(lbG));
      }
      sp += __builtin_popcountll(others);
      ts.c[3] += uint32_t(__builtin_popcountll(others));
      __builtin_amdgcn_wave_barrier();
      level = cl;
      node = first + uint32_t(jn);
      have = true;
    }
  }
}

__device__ __forceinline__ void flush_stats(const TraverseStats& ts, unsigned long long* g) {
  if (g != nullptr && (threadIdx.x & (WAVE - 1)) == 0) {
#pragma unroll
    for (int i = 0; i < 5; ++i) atomicAdd(g + i, (unsigned long long)ts.c[i]);
  }
}

// XCD-aware mapping of (block, wave, iteration) -> query group: blocks that share an XCD (and its
// L2) work on one contiguous window of Morton-ordered groups at a time.
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
  // local group index g (0 <= g < groups_per_xcd) -> global group
  __device__ __forceinline__ uint32_t global(uint32_t g) const { return xcd_first + g; }
};

}  // namespace pclhip
