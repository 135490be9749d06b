// lane_search.hpp -- exact 1-NN with ONE LANE PER QUERY over the index's quad levels (pclhip_internal.hpp: LaneTree).
//
// traverse() (traverse.hpp) walks the tree with a whole wavefront: 64 queries share every node scan, and the wave pays
// for the union of what its lanes need -- measured in round 4 at 20 of 64 lanes busy per evaluation round and half of
// the wave's cycles waiting.  A query that comes with a SEED next to its answer (the previous iteration's match) needs
// almost none of that: evaluate the seed's own 16-point leaf, then ask whether the ball (query, best distance) lies
// strictly inside the kd CELL of that leaf -- if so no other point of the index can be nearer or tie (LaneTree::qcell)
// and the query is done after ~100 instructions of its own lane.  Otherwise walk UP the quad levels (64, 256, 1024 ...
// points) to the first node whose cell holds the ball, and search that node top-down by the exact box bound
// (point_box_lb: bit-monotone in the distance's own operation order, as everywhere).  Nothing here talks to another
// lane: no LDS, no ballots, 8 waves per SIMD.
//
// What stays exact, and why:
//   * inside the node: every leaf whose box bound is <= the current best is evaluated (ties included), the bound only
//     shrinks, so a leaf skipped once stays skipped;
//   * outside the node: the cell test is STRICT, and every face of a cell is the extreme coordinate of the points on
//     the other side (quad_cell_kernel), so for a point p outside and a face F along axis a between it and the query,
//     |fl(q_a - p_a)| >= |fl(q_a - F)| (rounding is monotone) and fl(d^2(q, p)) >= fl(fl(q_a - p_a)^2) >= fl(g^2) > best.
// Reference: FLANN KDTreeSingleIndex::findNeighbors as called by KdTreeFLANN::nearestKSearch
// (kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp:131-135) -- same results, ties to the lower index.
#pragma once

#include "traverse.hpp"

namespace pclhip {

// the ball (q, r2) lies strictly inside the cell: the nearest face is more than sqrt(r2) away, decided on squares
__device__ __forceinline__ bool ball_inside_cell(const Box& c, float qx, float qy, float qz, float r2) {
  const float gx = fminf(__fsub_rn(qx, c.lo.x), __fsub_rn(c.hi.x, qx));
  const float gy = fminf(__fsub_rn(qy, c.lo.y), __fsub_rn(c.hi.y, qy));
  const float gz = fminf(__fsub_rn(qz, c.lo.z), __fsub_rn(c.hi.z, qz));
  const float g = fminf(gx, fminf(gy, gz));
  return g > 0.0f && __fmul_rn(g, g) > r2;
}

// A leaf near the query without any seed: down the quad levels, at every node into the child whose box is nearest
// (the trip count is the tree's height: uniform over the wave).  Any leaf makes a valid start -- it only gives the
// search a radius.
__device__ __forceinline__ uint32_t lane_greedy_leaf(const LaneTree& lt, float qx, float qy, float qz) {
  uint32_t off = 0;
  for (int q = 0; q < lt.top; ++q) off += lane_tree_count(lt.nleaf, q);
  uint32_t node = 0;
  for (int q = lt.top; q >= 1; --q) {
    const uint32_t cc = lane_tree_count(lt.nleaf, q - 1);
    off -= cc;
    float best = __builtin_inff();
    uint32_t bc = 0;
#pragma unroll
    for (uint32_t c = 0; c < 4u; ++c) {
      const uint32_t ci = 4u * node + c;
      if (ci < cc) {
        const Box b = lt.qbox[off + ci];
        const float lb = point_box_lb(qx, qy, qz, b.lo.x, b.lo.y, b.lo.z, b.hi.x, b.hi.y, b.hi.z);
        if (lb < best) {
          best = lb;
          bc = c;
        }
      }
    }
    node = 4u * node + bc;
  }
  return node;
}

// The search of one lane.  `pol` (NN1MinT<1> or NN1) has evaluated leaf `home` already.  Returns false when the ball
// does not fit a node of level <= max_up (the caller hands the query to the pass that has no such cap); `pol` then holds
// a valid upper bound.
template <class Pol>
__device__ __forceinline__ bool lane_search(const LaneTree& lt, const float* __restrict__ soa, float qx, float qy, float qz,
                                            Pol& pol, uint32_t home, int max_up) {
  const float qxa[1] = {qx}, qya[1] = {qy}, qza[1] = {qz};
  int q = 0;
  uint32_t node = home, off = 0, cnt = lt.nleaf;
  for (;;) {  // up: the first node whose cell holds the ball (the root's is all of space)
    if (q == lt.top) break;
    const Box c = lt.qcell[off + node];
    if (ball_inside_cell(c, qx, qy, qz, pol.worst(0))) break;
    if (q >= max_up) return false;
    off += cnt;
    cnt = (cnt + 3u) >> 2;
    node >>= 2;
    ++q;
  }
  if (q == 0) return true;
  // down: depth first, the children still to visit as four bits per level
  const int q0 = q;
  uint64_t pend = 0;
  bool down = true;
  for (;;) {
    if (down) {
      const uint32_t cc = lane_tree_count(lt.nleaf, q - 1);
      const uint32_t coff = off - cc;
      float lb[4];
      uint32_t m = 0;
#pragma unroll
      for (uint32_t c = 0; c < 4u; ++c) {
        const uint32_t ci = 4u * node + c;
        lb[c] = __builtin_inff();
        if (ci < cc) {
          const Box b = lt.qbox[coff + ci];
          lb[c] = point_box_lb(qx, qy, qz, b.lo.x, b.lo.y, b.lo.z, b.hi.x, b.hi.y, b.hi.z);
          if (lb[c] <= pol.worst(0) && !(q == 1 && ci == home)) m |= 1u << c;
        }
      }
      if (q == 1) {  // the children are leaves: evaluate them here, each against the bound of its moment
        while (m != 0u) {
          const uint32_t c = uint32_t(__builtin_ctz(m));
          m &= m - 1u;
          const float l = c == 0u ? lb[0] : (c == 1u ? lb[1] : (c == 2u ? lb[2] : lb[3]));
          if (l <= pol.worst(0)) pol.leaf_global(soa, 4u * node + c, qxa, qya, qza);
        }
      }
      pend |= uint64_t(m) << (4 * q);
      down = false;
    }
    const uint32_t m = uint32_t(pend >> (4 * q)) & 15u;
    if (m == 0u) {
      if (q == q0) break;
      off += lane_tree_count(lt.nleaf, q);
      node >>= 2;
      ++q;
      continue;
    }
    const uint32_t c = uint32_t(__builtin_ctz(m));
    pend &= ~(uint64_t(1) << (4 * q + int(c)));
    off -= lane_tree_count(lt.nleaf, q - 1);
    node = 4u * node + c;
    --q;
    down = true;
  }
  return true;
}

}  // namespace pclhip
