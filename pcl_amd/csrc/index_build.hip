// index_build.hip -- GPU construction of the kd-ordered implicit wide BVH.
//
//   finite/selected records -> float4 (w = original index) -> kd order (kd_order below): top rounds by radix
//   selection + partition (kp_*), bottom rounds inside one workgroup's LDS (kd_block_kernel) -> leaf boxes and
//   discs (16 points) -> 64-ary box levels.
//
// Replaces the build half of pcl::KdTreeFLANN<PointT>::setInputCloud
// (kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp:99-136,428-498): non-finite points are
// dropped, an optional index list selects a subset, results refer to original cloud indices.
// Hand-written throughout (no rocprim): the sorting variants of rounds 1 and 2 (a global radix sort per kd round, a
// Morton order) lived here behind environment switches for A/B runs and were removed in round 3 -- DESIGN.md section 3
// keeps their measurements.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string.h>


#include <cfloat>
#include <cstdlib>
#include <cmath>
#include <vector>

#include "pclhip_internal.hpp"
#include "device_scan.hpp"

namespace pclhip {

namespace {

__device__ __forceinline__ const float* record(const void* base, size_t stride, uint64_t i) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + i * stride);
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// one 16-lane group per leaf
// One 16-lane group per leaf, one pass over the points: the leaf's box, and its structure-of-arrays copy
// x[16] y[16] z[16] w[16] (w = original index bits, 256 B).  Pad slots of the last leaf get +FLT_MAX sentinels with index
// 0xFFFFFFFF in the copy (the points array itself may end at n) and do not count for the box.
__global__ __launch_bounds__(256) void leaf_box_soa_kernel(const float4* __restrict__ pts, uint32_t n, uint32_t nleaf,
                                                           Box* __restrict__ box, float* __restrict__ soa) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t leaf = i / LEAF;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  float4 p = make_float4(FLT_MAX, FLT_MAX, FLT_MAX, __uint_as_float(0xFFFFFFFFu));
  if (i < n) {
    p = pts[i];
    lo[0] = hi[0] = p.x; lo[1] = hi[1] = p.y; lo[2] = hi[2] = p.z;
  }
  if (leaf < nleaf) {
    float* l = soa + size_t(leaf) * (4 * LEAF) + (i % LEAF);
    l[0] = p.x;
    l[LEAF] = p.y;
    l[2 * LEAF] = p.z;
    l[3 * LEAF] = p.w;
  }
#pragma unroll
  for (int o = LEAF / 2; o > 0; o >>= 1) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      lo[d] = fminf(lo[d], __shfl_xor(lo[d], o));
      hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o));
    }
  }
  if ((i % LEAF) == 0 && leaf < nleaf) {
    Box b;
    b.lo = make_float4(lo[0], lo[1], lo[2], 0.0f);
    b.hi = make_float4(hi[0], hi[1], hi[2], 0.0f);
    box[leaf] = b;
  }
}

// One thread per leaf: the bounded cylinder ("disc") of its points (traverse.hpp: point_disc_lb) -- centre c
// (their mean, as a float), radius R >= |p - c|, direction n of least variance (double covariance, cyclic
// Jacobi) shrunk so that the stored float vector has |n|^2 >= 1 - 1e-6 and |n| <= 1, half thickness
// hn >= |n.(p - c)|.  R and hn are taken in double with the STORED float c and n and rounded upwards, so the
// disc holds every point of the leaf whatever the quality of the fit.  Anything non-finite -> the disc that
// bounds nothing (R = hn = FLT_MAX).
// cyclic Jacobi on a symmetric 3x3 in float, a fixed number of sweeps (see leaf_disc_kernel)
__device__ __forceinline__ void jacobi_eig3_f32(float A[3][3], float V[3][3], float w[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? 1.0f : 0.0f;
  for (int sweep = 0; sweep < 5; ++sweep) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        const float apq = A[p][q];
        if (fabsf(apq) < 1e-30f) continue;
        const float theta = (A[q][q] - A[p][p]) / (2.0f * apq);
        const float t = (theta >= 0.0f ? 1.0f : -1.0f) / (fabsf(theta) + sqrtf(theta * theta + 1.0f));
        const float c = 1.0f / sqrtf(t * t + 1.0f), s = t * c;
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // A <- A * J
          const float akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // A <- J^T * A
          const float apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) w[i] = A[i][i];
}

__global__ __launch_bounds__(256) void leaf_disc_kernel(const float4* __restrict__ pts, uint32_t n, uint32_t nleaf,
                                                        float4* __restrict__ disc) {
  const uint32_t leaf = blockIdx.x * blockDim.x + threadIdx.x;
  if (leaf >= nleaf) return;
  const uint32_t b = leaf * LEAF;
  const int m = int(b < n ? (n - b < uint32_t(LEAF) ? n - b : uint32_t(LEAF)) : 0u);  // real points (the rest is padding)
  float4 cR = make_float4(0, 0, 0, FLT_MAX), nh = make_float4(0, 0, 0, FLT_MAX);
  if (m > 0) {
    double mean[3] = {0, 0, 0};
    float4 p[LEAF];
#pragma unroll
    for (int i = 0; i < LEAF; ++i) {
      p[i] = pts[b + (i < m ? i : 0)];
      if (i < m) { mean[0] += p[i].x; mean[1] += p[i].y; mean[2] += p[i].z; }
    }
    const float c[3] = {float(mean[0] / m), float(mean[1] / m), float(mean[2] / m)};
    double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    double r2max = 0.0;
#pragma unroll
    for (int i = 0; i < LEAF; ++i) {
      if (i < m) {
        const double d[3] = {double(p[i].x) - c[0], double(p[i].y) - c[1], double(p[i].z) - c[2]};
        for (int r = 0; r < 3; ++r)
          for (int cc = r; cc < 3; ++cc) A[r][cc] += d[r] * d[cc];
        r2max = fmax(r2max, d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      }
    }
    A[1][0] = A[0][1]; A[2][0] = A[0][2]; A[2][1] = A[1][2];
    // direction of least variance.  ANY direction makes a valid disc (R and hn below are measured for the direction
    // that comes out), a better one only makes it thinner: five Jacobi sweeps in float on the trace-normalised
    // matrix are plenty (the double-precision solver of closed_forms.hpp run to exhaustion cost 0.66 ms of a
    // 3.8 ms build of 10M points)
    float V[3][3], w[3];
    {
      const double tr = A[0][0] + A[1][1] + A[2][2];
      const double sc = tr > 0.0 ? 1.0 / tr : 0.0;
      float B[3][3];
      for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) B[r][cc] = float(A[r][cc] * sc);
      jacobi_eig3_f32(B, V, w);
    }
    const int k = (w[0] <= w[1] && w[0] <= w[2]) ? 0 : ((w[1] <= w[2]) ? 1 : 2);
    const double nd[3] = {double(V[0][k]), double(V[1][k]), double(V[2][k])};
    const double len = sqrt(nd[0] * nd[0] + nd[1] * nd[1] + nd[2] * nd[2]);
    float nf[3] = {0, 0, 0};
    bool ok = len > 0.5 && len < 2.0;
    if (ok) {
      for (int d = 0; d < 3; ++d) nf[d] = float(nd[d] / len * (1.0 - 4e-7));
      const double nn = double(nf[0]) * nf[0] + double(nf[1]) * nf[1] + double(nf[2]) * nf[2];
      ok = nn <= 1.0 && nn >= 1.0 - 1e-6;
    }
    if (ok) {
      double hmax = 0.0;
#pragma unroll
      for (int i = 0; i < LEAF; ++i) {
        if (i < m) {
          const double s = double(nf[0]) * (double(p[i].x) - c[0]) + double(nf[1]) * (double(p[i].y) - c[1]) +
                           double(nf[2]) * (double(p[i].z) - c[2]);
          hmax = fmax(hmax, fabs(s));
        }
      }
      float R = float(sqrt(r2max) * (1.0 + 1e-6)), hn = float(hmax * (1.0 + 1e-6));
      R = nextafterf(R, INFINITY);
      hn = nextafterf(hn, INFINITY);
      if (isfinite(R) && isfinite(hn) && isfinite(c[0]) && isfinite(c[1]) && isfinite(c[2])) {
        cR = make_float4(c[0], c[1], c[2], R);
        nh = make_float4(nf[0], nf[1], nf[2], hn);
      }
    }
  }
  disc[2 * leaf] = cR;
  disc[2 * leaf + 1] = nh;
}

// per-block sums of the squared diagonals of the leaf boxes (fixed order: the mean is reproducible)
// part[b] = sum of squared leaf diagonals; part[nb + b] / part[2 nb + b] = sums of the discs' half thickness / radius
// (how thin the leaves are against their width: what the disc bounds live on)
__global__ __launch_bounds__(256) void leaf_diag_kernel(const Box* __restrict__ box, const float4* __restrict__ disc,
                                                        uint32_t nleaf, double* __restrict__ part) {
  __shared__ double red[3][256];
  double a = 0.0, hsum = 0.0, rsum = 0.0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nleaf; i += gridDim.x * blockDim.x) {
    const Box b = box[i];
    const double ex = double(b.hi.x) - b.lo.x, ey = double(b.hi.y) - b.lo.y, ez = double(b.hi.z) - b.lo.z;
    const double d2 = ex * ex + ey * ey + ez * ez;
    a += (d2 < 1e30) ? d2 : 0.0;  // the padded last leaf holds sentinels
    const float R = disc[2 * size_t(i)].w, hn = disc[2 * size_t(i) + 1].w;
    if (R < 1e30f && hn < 1e30f) {  // degenerate discs carry FLT_MAX
      hsum += double(hn);
      rsum += double(R);
    }
  }
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = hsum;
  red[2][threadIdx.x] = rsum;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (int(threadIdx.x) < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
      red[2][threadIdx.x] += red[2][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[blockIdx.x] = red[0][0];
    part[gridDim.x + blockIdx.x] = red[1][0];
    part[2 * gridDim.x + blockIdx.x] = red[2][0];
  }
}

// one wavefront per parent node
__global__ __launch_bounds__(256) void node_box_kernel(const Box* child, uint32_t nchild, Box* parent, uint32_t nparent) {
  const uint32_t node = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const uint32_t lane = threadIdx.x & 63;
  if (node >= nparent) return;
  const uint32_t c = node * FANOUT + lane;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  if (c < nchild) {
    const Box b = child[c];
    lo[0] = b.lo.x; lo[1] = b.lo.y; lo[2] = b.lo.z;
    hi[0] = b.hi.x; hi[1] = b.hi.y; hi[2] = b.hi.z;
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    lo[d] = wave_min(lo[d]);
    hi[d] = wave_max(hi[d]);
  }
  if (lane == 0) {
    Box b;
    b.lo = make_float4(lo[0], lo[1], lo[2], 0.0f);
    b.hi = make_float4(hi[0], hi[1], hi[2], 0.0f);
    parent[node] = b;
  }
}


// ---- the per-lane search structure (pclhip_internal.hpp: LaneTree) ------------------------------------------------
// boxes of one quad level from the level below: one thread per parent, four consecutive children
__device__ __forceinline__ void quad_box_node(const Box* __restrict__ child, uint32_t nchild, Box* __restrict__ parent,
                                              uint32_t i) {
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
#pragma unroll
  for (uint32_t c = 0; c < 4u; ++c) {
    if (4u * i + c < nchild) {
      const Box b = child[4u * i + c];
      lo[0] = fminf(lo[0], b.lo.x); lo[1] = fminf(lo[1], b.lo.y); lo[2] = fminf(lo[2], b.lo.z);
      hi[0] = fmaxf(hi[0], b.hi.x); hi[1] = fmaxf(hi[1], b.hi.y); hi[2] = fmaxf(hi[2], b.hi.z);
    }
  }
  Box b;
  b.lo = make_float4(lo[0], lo[1], lo[2], 0.0f);
  b.hi = make_float4(hi[0], hi[1], hi[2], 0.0f);
  parent[i] = b;
}
__global__ __launch_bounds__(256) void quad_box_kernel(const Box* __restrict__ child, uint32_t nchild, Box* __restrict__ parent,
                                                       uint32_t nparent) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nparent) quad_box_node(child, nchild, parent, i);
}

struct B3 {
  float lo[3], hi[3];
};
__device__ __forceinline__ bool b3_empty(const B3& b) { return !(b.lo[0] <= b.hi[0]); }
__device__ __forceinline__ B3 b3_union(const B3& a, const B3& b) {
  B3 r;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    r.lo[d] = fminf(a.lo[d], b.lo[d]);
    r.hi[d] = fmaxf(a.hi[d], b.hi[d]);
  }
  return r;
}
// X lies in front of Y in the kd order.  Both non-empty: an axis along which every point of X is <= every point of Y
// (the widest gap if several do) puts a face into both cells: X ends where Y's points begin, Y begins where X's points
// end -- each face is set by the OTHER side's extreme, so that "strictly inside my cell" means "strictly in front of
// every point of the other side".  No such axis (the order is not the kd partition this structure assumes): both
// cells are inverted and contain nothing, which sends every query of the two subtrees up to the parent.
__device__ __forceinline__ void cell_split(const B3& X, const B3& Y, B3& cx, B3& cy) {
  if (b3_empty(X) || b3_empty(Y)) return;
  int axis = -1;
  float gap = -1.0f;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float g = Y.lo[d] - X.hi[d];
    if (X.hi[d] <= Y.lo[d] && g > gap) {
      gap = g;
      axis = d;
    }
  }
  if (axis < 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      cx.lo[d] = cy.lo[d] = INFINITY;
      cx.hi[d] = cy.hi[d] = -INFINITY;
    }
    return;
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (d == axis) {
      cx.hi[d] = fminf(cx.hi[d], Y.lo[d]);
      cy.lo[d] = fmaxf(cy.lo[d], X.hi[d]);
    }
  }
}
// cells of one quad level from the cells of the level above and the boxes of the level itself: one thread per PARENT.
// The four children are two halves of two (a four-way cut along one axis is the same thing seen as binary cuts).
__device__ __forceinline__ void quad_cell_node(const Box* __restrict__ pcell, int parent_is_root, const Box* __restrict__ cbox,
                                               uint32_t nchild, Box* __restrict__ ccell, uint32_t i) {
  B3 pc;
  if (parent_is_root) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      pc.lo[d] = -INFINITY;
      pc.hi[d] = INFINITY;
    }
  } else {
    const Box b = pcell[i];
    pc.lo[0] = b.lo.x; pc.lo[1] = b.lo.y; pc.lo[2] = b.lo.z;
    pc.hi[0] = b.hi.x; pc.hi[1] = b.hi.y; pc.hi[2] = b.hi.z;
  }
  B3 cb[4], cc[4];
#pragma unroll
  for (uint32_t c = 0; c < 4u; ++c) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      cb[c].lo[d] = FLT_MAX;
      cb[c].hi[d] = -FLT_MAX;
    }
    if (4u * i + c < nchild) {
      const Box b = cbox[4u * i + c];
      cb[c].lo[0] = b.lo.x; cb[c].lo[1] = b.lo.y; cb[c].lo[2] = b.lo.z;
      cb[c].hi[0] = b.hi.x; cb[c].hi[1] = b.hi.y; cb[c].hi[2] = b.hi.z;
    }
    cc[c] = pc;
  }
  {
    const B3 A = b3_union(cb[0], cb[1]), Bh = b3_union(cb[2], cb[3]);
    B3 ca = pc, cbh = pc;
    cell_split(A, Bh, ca, cbh);
    cc[0] = cc[1] = ca;
    cc[2] = cc[3] = cbh;
  }
  cell_split(cb[0], cb[1], cc[0], cc[1]);
  cell_split(cb[2], cb[3], cc[2], cc[3]);
#pragma unroll
  for (uint32_t c = 0; c < 4u; ++c) {
    if (4u * i + c < nchild) {
      Box b;
      b.lo = make_float4(cc[c].lo[0], cc[c].lo[1], cc[c].lo[2], 0.0f);
      b.hi = make_float4(cc[c].hi[0], cc[c].hi[1], cc[c].hi[2], 0.0f);
      ccell[4u * i + c] = b;
    }
  }
}
__global__ __launch_bounds__(256) void quad_cell_kernel(const Box* __restrict__ pcell, uint32_t nparent, int parent_is_root,
                                                        const Box* __restrict__ cbox, uint32_t nchild,
                                                        Box* __restrict__ ccell) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nparent) quad_cell_node(pcell, parent_is_root, cbox, nchild, ccell, i);
}
// The SMALL quad levels in one launch (a 10M-point index has eleven levels; from the fifth up they hold 2,442, 611, 153, ...
// nodes: seven box launches and seven cell launches of a few microseconds of work and ~5 us of launch each).  One workgroup:
// boxes of the levels [q_from, q_top] bottom-up, then the cells of the levels [q_from - 1, q_top - 1] top-down, a barrier
// between levels (the levels' arrays are global memory written and read by this one workgroup: a WORKGROUP-scope fence orders
// them -- its waves share one vector cache; the device-scope fence that stood here until late in round 6 wrote the XCD's
// L2 back fourteen times: 104 -> 62 us at 10M points).  qbox / qcell: the arrays of all levels, level q at offset
// off(q) = sum of the counts below it.
__global__ __launch_bounds__(1024) void quad_top_kernel(Box* __restrict__ qbox, Box* __restrict__ qcell, uint32_t nleaf, int q_from,
                                                        int q_top) {
  const auto count = [&](int q) { return lane_tree_count(nleaf, q); };
  size_t off = 0;
  for (int q = 0; q < q_from - 1; ++q) off += count(q);
  // off = offset of level q_from - 1 (the children of the first level built here)
  size_t o = off;
  for (int q = q_from; q <= q_top; ++q) {
    const uint32_t cc = count(q - 1), cp = count(q);
    for (uint32_t i = threadIdx.x; i < cp; i += blockDim.x) quad_box_node(qbox + o, cc, qbox + o + cc, i);
    o += cc;
    __threadfence_block();
    __syncthreads();
  }
  // o = offset of level q_top
  for (int q = q_top; q >= q_from; --q) {
    const uint32_t cc = count(q - 1), cp = count(q);
    const size_t coff = o - cc;
    for (uint32_t i = threadIdx.x; i < cp; i += blockDim.x)
      quad_cell_node(qcell + o, q == q_top ? 1 : 0, qbox + coff, cc, qcell + coff, i);
    o = coff;
    __threadfence_block();
    __syncthreads();
  }
}

}  // namespace

namespace {

constexpr int KD_CHUNK_MAX = 4096;

// one wavefront per chunk of `chunk` consecutive points -> chunk box
__global__ __launch_bounds__(256) void kd_chunk_box_kernel(const float4* __restrict__ pts, uint32_t n, uint32_t chunk,
                                                           uint32_t nchunks, Box* __restrict__ out) {
  const uint32_t c = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const uint32_t lane = threadIdx.x & 63;
  if (c >= nchunks) return;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  const uint64_t b = uint64_t(c) * chunk;
  uint64_t e = b + chunk;
  if (e > n) e = n;
  for (uint64_t i = b + lane; i < e; i += WAVE) {
    const float4 p = pts[i];
    lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
    hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    lo[d] = wave_min(lo[d]);
    hi[d] = wave_max(hi[d]);
  }
  if (lane == 0) {
    Box bx;
    bx.lo = make_float4(lo[0], lo[1], lo[2], 0.0f);
    bx.hi = make_float4(hi[0], hi[1], hi[2], 0.0f);
    out[c] = bx;
  }
}


__device__ __forceinline__ uint32_t orderable(float f) {
  const uint32_t u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}



// ---- bottom kd rounds in one workgroup ------------------------------------------------------------
// Segments of <= 4096 points are ordered entirely inside LDS: the rounds that cut a 4096-point segment into four
// 1024-point slabs and those into 256-point slabs, then four BINARY cuts (256 -> 128 -> 64 -> 32 -> 16, each along the
// widest axis of the piece it cuts) run back to back in one launch.  The points stay where they are (x, y, z planes in
// LDS); a 16-bit permutation moves.  Per level: bounding box of every sub-segment -> widest axis -> keys = exact
// unsigned order of the coordinate along it, then
//   * the two four-way levels (sub-segments of 4096 / 1024 points) only need every sub-segment CUT at its quartile
//     order statistics -- the order inside a slab is irrelevant, the next level re-orders it along another axis.  So
//     they do what the top rounds do across the grid (kp_* below), inside LDS: the three quartile keys by radix
//     selection (7-bit digits from the top of the key range of the sub-segment: per-sub-segment histograms with packed
//     16-bit counters, after the first pass only the keys inside the three selected bins count), every point classified
//     against them (< s1, = s1, between, ..., > s3: seven classes, monotone in the key), and the permutation rearranged
//     by (class, previous position) with a prefix sum over the threads -- deterministic; the positional slab boundaries
//     fall inside the "= s_k" classes, so ties at a splitter are split by count.  (Until round 4 these levels were full
//     bitonic sorts of (key, position) pairs.)
//   * the four binary levels (256, 128, 64 and 32 points) are bitonic sorts inside one wavefront (a thread holds four
//     consecutive positions, so a wavefront holds a whole 256-point sub-segment): shuffles only.
// Why binary below 1024: four slabs along ONE axis make a square cell into four 4:1 strips.  Until the end of round 4 the
// 256-point level was such a cut, so a 64-point cell -- one wavefront of queries, and the run the stand-off search seeds --
// was a 16 x 4 strip of point spacings on a surface; two binary cuts make it an 8 x 8 patch (and the 16-point leaves
// 4 x 4 as before).  A strip's neighbourhood is a third longer than a square's: every search launch got 6-18 % faster
// (`ms_per_step` 1.088 -> 0.991 at 10M points, cold launch 1.83 -> 1.50 ms, same call and box) for 0.13 ms more in this
// kernel (the 256- and 128-point sorts: 64 more compare-exchange stages).
// Keys are exact float orders, so cells keep disjoint interiors.
#ifndef PCLHIP_KDB_MINW
#define PCLHIP_KDB_MINW 8     // waves per SIMD the register budget admits: 8 = 64 VGPRs = TWO 16-wave workgroups per CU.  The
                              // kernel waits on LDS round trips (shuffles, gathers through the permutation), not on VALU
                              // issue: 917 -> 801 us with the packed keys, -> 584 us with the second workgroup resident
                              // (10M points, round 6; 6 VGPRs spill on the cold 64-bit path)
#endif
#ifndef PCLHIP_KDB_PACKED
#define PCLHIP_KDB_PACKED 1   // A/B: 0 = the (key, position) 64-bit compare-exchange in every binary level
#endif
constexpr int KDB_N = 4096;
constexpr int KDB_THREADS = 1024;
constexpr int KDB_WAVES = KDB_THREADS / WAVE;
constexpr int KDB_FOURWAY_FROM = 1024;   // sub-segments of this size and above are cut four ways, smaller ones in two
constexpr int KDB_MAXSUB = KDB_N / KDB_FOURWAY_FROM;  // sub-segments of the smallest four-way level
struct KdBlockLds {
  float x[KDB_N], y[KDB_N], z[KDB_N];
  uint16_t perm[KDB_N];
  float wbox[KDB_WAVES][6];
  uint32_t hist[KDB_MAXSUB][3][64];   // 128 bins of 16 bits per (sub-segment, splitter): 12 KB -- with the 56 KB of points
                                      // and permutation two blocks still fit a CU's LDS (256 bins: one block, measured)
  uint32_t sel_prefix[KDB_MAXSUB][3], sel_rank[KDB_MAXSUB][3], sel_digit[KDB_MAXSUB][3];
  uint32_t max_bits;                  // widest key range among the block's sub-segments at this level
  uint32_t wscan[KDB_WAVES][4];       // per-wave class totals: seven 16-bit fields
};

#ifdef PCLHIP_KDB_TICKS  // timing probe (scratch/kdb_ticks.py): clock64 ticks of thread 0 per phase, summed over the blocks
__device__ unsigned long long g_kdb_ticks[8];
#define KDB_LAP(i)                                            \
  do {                                                        \
    if (threadIdx.x == 0) {                                   \
      const unsigned long long kdb_now = clock64();           \
      kdb_acc[i] += kdb_now - kdb_t;                          \
      kdb_t = kdb_now;                                        \
    }                                                         \
  } while (0)
#else
#define KDB_LAP(i) (void)0
#endif
__global__ __launch_bounds__(KDB_THREADS, PCLHIP_KDB_MINW) void kd_block_kernel(const float4* __restrict__ in, uint32_t n,
                                                               float4* __restrict__ out, uint32_t top_nsub,
                                                               uint32_t bottom_nsub, uint32_t* __restrict__ rank) {
  __shared__ KdBlockLds s;
  const uint32_t t = threadIdx.x;
  const uint32_t lane = t & 63u, wave = t >> 6;
  const uint32_t base = blockIdx.x * uint32_t(KDB_N);
  if (base >= n) return;
  const uint32_t cnt = (n - base) < uint32_t(KDB_N) ? (n - base) : uint32_t(KDB_N);
#ifdef PCLHIP_KDB_TICKS
  unsigned long long kdb_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, kdb_t = clock64();
#endif
  for (uint32_t p = t; p < uint32_t(KDB_N); p += KDB_THREADS) {
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (p < cnt) v = in[base + p];
    s.x[p] = v.x; s.y[p] = v.y; s.z[p] = v.z;
    s.perm[p] = uint16_t(p);
  }
  __syncthreads();
  KDB_LAP(0);   // load
  for (uint32_t nsub = top_nsub; nsub >= bottom_nsub; nsub = (nsub >= uint32_t(KDB_FOURWAY_FROM)) ? nsub / 4u : nsub / 2u) {
    // (a) bounding box of the sub-segment this thread's four positions belong to
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    uint32_t idx[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t p = 4u * t + uint32_t(e);
      idx[e] = s.perm[p];
      if (p < cnt) {  // the points of a sub-segment always precede its padding (padding sorts last)
        const float px = s.x[idx[e]], py = s.y[idx[e]], pz = s.z[idx[e]];
        lo[0] = fminf(lo[0], px); lo[1] = fminf(lo[1], py); lo[2] = fminf(lo[2], pz);
        hi[0] = fmaxf(hi[0], px); hi[1] = fmaxf(hi[1], py); hi[2] = fmaxf(hi[2], pz);
      }
    }
    const uint32_t group = nsub / 4u;  // threads per sub-segment: 8 ... 1024
    const uint32_t inwave = group < 64u ? group : 64u;
    for (uint32_t o = inwave >> 1; o > 0; o >>= 1) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        lo[d] = fminf(lo[d], __shfl_xor(lo[d], int(o)));
        hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], int(o)));
      }
    }
    if (group > 64u) {  // wave-uniform: the sub-segment spans several wavefronts
      if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          s.wbox[wave][d] = lo[d];
          s.wbox[wave][3 + d] = hi[d];
        }
      }
      __syncthreads();
      const uint32_t wpg = group / 64u, w0 = (wave / wpg) * wpg;
      for (uint32_t w = w0; w < w0 + wpg; ++w) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          lo[d] = fminf(lo[d], s.wbox[w][d]);
          hi[d] = fmaxf(hi[d], s.wbox[w][3 + d]);
        }
      }
    }
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    int a = 0;
    float best = ex;
    if (ey > best) { best = ey; a = 1; }
    if (ez > best) { a = 2; }
    uint32_t kk[4], pp[4];
    KDB_LAP(1);   // boxes + axis (all levels)
    if (nsub >= uint32_t(KDB_FOURWAY_FROM)) {
      // ---- four-way level: quartile selection + partition (block-uniform branch) ---------------------------------
      const uint32_t sub = t / group;
      const bool empty = !(lo[0] <= hi[0]);  // a sub-segment of padding only
      const uint32_t kmin = orderable(a == 0 ? lo[0] : (a == 1 ? lo[1] : lo[2]));
      const uint32_t W = empty ? 0u : orderable(a == 0 ? hi[0] : (a == 1 ? hi[1] : hi[2])) - kmin;
      const uint32_t PAD = W + 1u;  // above every real key (finite coordinates: W < 2^32 - 1)
      const int nb = 32 - __builtin_clz(PAD);  // bits of the key range, >= 1
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t p = 4u * t + uint32_t(e);
        const float c = a == 0 ? s.x[idx[e]] : (a == 1 ? s.y[idx[e]] : s.z[idx[e]]);
        kk[e] = p < cnt ? orderable(c) - kmin : PAD;
        pp[e] = idx[e];
      }
      const uint32_t tin = t - sub * group;  // thread inside its sub-segment
      if (tin < 3u) {
        s.sel_prefix[sub][tin] = 0u;
        s.sel_rank[sub][tin] = (tin + 1u) * group;  // position (k + 1) * nsub / 4 of the sorted order
      }
      if (t == 0u) s.max_bits = 0u;
      __syncthreads();
      if (tin == 0u) atomicMax(&s.max_bits, uint32_t(nb));
      __syncthreads();
      const int npass = int(s.max_bits + 6u) / 7;
      KDB_LAP(2);   // four-way: keys + setup
#ifdef PCLHIP_KDB_TICKS
      if (threadIdx.x == 0) kdb_acc[7] += (unsigned long long)npass;
#endif
      for (int pass = 0; pass < npass; ++pass) {
        const int hi_b = nb - 7 * pass;  // this pass decides bits [lo_b, hi_b) of the sub-segment's keys
        const bool active = hi_b > 0;
        const int lo_b = hi_b > 7 ? hi_b - 7 : 0;
        const uint32_t dmask = active ? ((1u << (hi_b - lo_b)) - 1u) : 0u;
        for (uint32_t i = t; i < uint32_t(KDB_MAXSUB * 3 * 64); i += KDB_THREADS) (&s.hist[0][0][0])[i] = 0u;
        __syncthreads();
        if (active) {
          const uint32_t p0 = s.sel_prefix[sub][0], p1 = s.sel_prefix[sub][1], p2 = s.sel_prefix[sub][2];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t K = kk[e], dg = (K >> lo_b) & dmask, add = 1u << (16u * (dg & 1u));
            if (pass == 0) {  // no bits decided yet: one histogram serves the three splitters
              atomicAdd(&s.hist[sub][0][dg >> 1], add);
            } else {          // (hi_b <= 25 here)
              const uint32_t top = K >> hi_b;
              if (top == (p0 >> hi_b)) atomicAdd(&s.hist[sub][0][dg >> 1], add);
              if (top == (p1 >> hi_b)) atomicAdd(&s.hist[sub][1][dg >> 1], add);
              if (top == (p2 >> hi_b)) atomicAdd(&s.hist[sub][2][dg >> 1], add);
            }
          }
        }
        __syncthreads();
        // one wavefront per (sub-segment, splitter): the bin that holds the rank
        const uint32_t nsubseg = uint32_t(KDB_N) / nsub;
        for (uint32_t pr = wave; pr < nsubseg * 3u; pr += uint32_t(KDB_WAVES)) {
          const uint32_t sb = pr / 3u, k = pr - sb * 3u;
          // (a sub-segment whose key range is exhausted adds nothing to the histograms: marked by rank 0xFFFFFFFF)
          const uint32_t rank = s.sel_rank[sb][k];
          if (rank == 0xFFFFFFFFu) continue;
          const uint32_t* h = s.hist[sb][pass == 0 ? 0 : k];
          const uint32_t wv = h[lane];  // bins 2 * lane, 2 * lane + 1
          const uint32_t c0 = wv & 0xFFFFu, c1 = wv >> 16;
          const uint32_t sum = c0 + c1;
          uint32_t incl = sum;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            if (int(lane) >= o) incl += v;
          }
          const uint32_t total = __shfl(incl, 63);
          if (total == 0u) {  // this sub-segment took no part in the pass (its key range is exhausted)
            if (lane == 0) s.sel_rank[sb][k] = 0xFFFFFFFFu;
            continue;
          }
          const uint32_t excl = incl - sum;
          if (rank >= excl && rank < incl) {
            const bool second = rank >= excl + c0;
            const uint32_t dg = 2u * lane + (second ? 1u : 0u), left = rank - excl - (second ? c0 : 0u);
            s.sel_digit[sb][k] = dg;  // folded into the prefix below, by a thread that knows the sub-segment's shift
            s.sel_rank[sb][k] = left;
          }
        }
        __syncthreads();
        // the threads of a sub-segment know its shift: the first three fold the chosen digits into the prefixes
        if (active && tin < 3u) s.sel_prefix[sub][tin] |= s.sel_digit[sub][tin] << lo_b;
      }
      __syncthreads();
      KDB_LAP(3);   // four-way: selection passes
      const uint32_t s1 = s.sel_prefix[sub][0], s2 = s.sel_prefix[sub][1], s3 = s.sel_prefix[sub][2];
      // classes, monotone in the key; layout by (class, previous position)
      uint32_t cls[4], w0 = 0u, w1 = 0u, w2 = 0u, w3 = 0u;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t K = kk[e];
        const uint32_t c = (K >= s1 ? 1u : 0u) + (K > s1 ? 1u : 0u) + (K >= s2 ? 1u : 0u) + (K > s2 ? 1u : 0u) +
                           (K >= s3 ? 1u : 0u) + (K > s3 ? 1u : 0u);
        cls[e] = c;
        const uint32_t add = 1u << (16u * (c & 1u));
        w0 += (c >> 1) == 0u ? add : 0u;
        w1 += (c >> 1) == 1u ? add : 0u;
        w2 += (c >> 1) == 2u ? add : 0u;
        w3 += (c >> 1) == 3u ? add : 0u;
      }
      uint32_t i0 = w0, i1 = w1, i2 = w2, i3 = w3;  // inclusive scan over the wavefront (inside one sub-segment)
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v0 = __shfl_up(i0, o), v1 = __shfl_up(i1, o), v2 = __shfl_up(i2, o), v3 = __shfl_up(i3, o);
        if (int(lane) >= o) { i0 += v0; i1 += v1; i2 += v2; i3 += v3; }
      }
      if (lane == 63u) { s.wscan[wave][0] = i0; s.wscan[wave][1] = i1; s.wscan[wave][2] = i2; s.wscan[wave][3] = i3; }
      __syncthreads();
      const uint32_t wpg = group / 64u, wfirst = (wave / wpg) * wpg;
      uint32_t b0 = 0u, b1 = 0u, b2 = 0u, b3 = 0u, t0 = 0u, t1 = 0u, t2 = 0u, t3 = 0u;  // waves before this one / all
      for (uint32_t w = wfirst; w < wfirst + wpg; ++w) {
        const uint32_t a0 = s.wscan[w][0], a1 = s.wscan[w][1], a2 = s.wscan[w][2], a3 = s.wscan[w][3];
        if (w < wave) { b0 += a0; b1 += a1; b2 += a2; b3 += a3; }
        t0 += a0; t1 += a1; t2 += a2; t3 += a3;
      }
      // exclusive prefix of this thread per class, class bases
      const uint32_t x0 = b0 + i0 - w0, x1 = b1 + i1 - w1, x2 = b2 + i2 - w2, x3 = b3 + i3 - w3;
      uint32_t cbase[7];
      {
        const uint32_t tot[7] = {t0 & 0xFFFFu, t0 >> 16, t1 & 0xFFFFu, t1 >> 16, t2 & 0xFFFFu, t2 >> 16, t3 & 0xFFFFu};
        uint32_t acc = 0u;
#pragma unroll
        for (int c = 0; c < 7; ++c) {
          cbase[c] = acc;
          acc += tot[c];
        }
      }
      uint32_t seen0 = 0u, seen1 = 0u, seen2 = 0u, seen3 = 0u;  // own earlier elements per class (packed like w*)
      uint32_t dest[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t c = cls[e], hsel = c >> 1, sh = 16u * (c & 1u);
        const uint32_t xw = hsel == 0u ? x0 : (hsel == 1u ? x1 : (hsel == 2u ? x2 : x3));
        const uint32_t sw = hsel == 0u ? seen0 : (hsel == 1u ? seen1 : (hsel == 2u ? seen2 : seen3));
        uint32_t cb = cbase[0];
#pragma unroll
        for (int q = 1; q < 7; ++q) cb = c == uint32_t(q) ? cbase[q] : cb;
        dest[e] = sub * nsub + cb + ((xw >> sh) & 0xFFFFu) + ((sw >> sh) & 0xFFFFu);
        const uint32_t add = 1u << sh;
        seen0 += hsel == 0u ? add : 0u;
        seen1 += hsel == 1u ? add : 0u;
        seen2 += hsel == 2u ? add : 0u;
        seen3 += hsel == 3u ? add : 0u;
      }
      __syncthreads();  // every reader of perm (this level's idx[]) is done
#pragma unroll
      for (int e = 0; e < 4; ++e) s.perm[dest[e]] = uint16_t(pp[e]);
      __syncthreads();
      KDB_LAP(4);   // four-way: classify + scan + permute
      continue;
    }
    // ---- binary levels (256 ... 32 points): bitonic sort inside a wavefront -------------------------------------
    // A wavefront holds a whole 256-point sub-segment (a thread four consecutive positions), so these levels touch no
    // other wavefront's slice of the permutation: wave barriers order their LDS traffic, no workgroup barrier.
    // PACKED form (round 6): the key of an element is its coordinate's unsigned order RELATIVE to the sub-segment's
    // minimum -- 24 bits or fewer for any cell that does not straddle zero or a dozen binades -- with the element's
    // position inside the sub-segment in the low 8 bits: ONE 32-bit word that is unique, orders by (coordinate, current
    // position), and tells where the winner of a slot came from.  A compare-exchange is then one shuffle, a min and a
    // max instead of two shuffles and a 64-bit compare with two selects (the 100 stages of these four levels were
    // 43 % of this kernel's cycles, VALU-issue-bound at 4 waves per SIMD: scratch/kdb_ticks.py).  The payload (the
    // index into the coordinate planes) is fetched once per level from the permutation slot the low bits name.
    // A wavefront with a wider sub-segment takes the (key, position) 64-bit form below.
    const uint32_t lp0 = (4u * t) & (nsub - 1u);   // position of the thread's first element inside its sub-segment
    const uint32_t segbase = (4u * t) & ~(nsub - 1u);
    bool packed = false;
#if PCLHIP_KDB_PACKED
    {
      const bool empty = !(lo[0] <= hi[0]);
      const uint32_t kmin = empty ? 0u : orderable(a == 0 ? lo[0] : (a == 1 ? lo[1] : lo[2]));
      const uint32_t W = empty ? 0u : orderable(a == 0 ? hi[0] : (a == 1 ? hi[1] : hi[2])) - kmin;
      packed = __builtin_amdgcn_ballot_w64(W >= 0x00FFFFFEu) == 0ull;   // wave-uniform: every sub-segment of the wave fits
      if (packed) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t p = 4u * t + uint32_t(e);
          const float c = a == 0 ? s.x[idx[e]] : (a == 1 ? s.y[idx[e]] : s.z[idx[e]]);
          // padding sorts last: above every real key (relative keys stay below 0xFFFFFE)
          kk[e] = (p < cnt ? ((orderable(c) - kmin) << 8) : 0xFFFFFF00u) | (lp0 + uint32_t(e));
        }
        for (uint32_t k = 2; k <= nsub; k <<= 1) {
          const bool up = (((4u * t) & (nsub - 1u)) & k) == 0u;   // k >= 4: the same for the thread's four elements
          for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            if (j >= 4u) {
              const bool take_min = (((4u * t) & j) == 0u) == up;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const uint32_t o = __shfl_xor(kk[e], int(j >> 2));
                kk[e] = take_min ? (kk[e] < o ? kk[e] : o) : (kk[e] > o ? kk[e] : o);
              }
            } else {
              uint32_t o[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = j == 1u ? kk[e ^ 1] : kk[e ^ 2];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const uint32_t i = 4u * t + uint32_t(e);
                const bool lower = (i & j) == 0u, upe = ((i & (nsub - 1u)) & k) == 0u;   // k == 2: differs inside the thread
                kk[e] = (lower == upe) ? (kk[e] < o[e] ? kk[e] : o[e]) : (kk[e] > o[e] ? kk[e] : o[e]);
              }
            }
          }
        }
        // slot 4t + e now holds the element that stood at position (kk & 255) of the sub-segment: its payload
#pragma unroll
        for (int e = 0; e < 4; ++e) pp[e] = s.perm[segbase + (kk[e] & 0xFFu)];
      }
    }
#endif
    if (!packed) {
      // (b) keys: the coordinate along that axis in unsigned order; padding sorts last.  The thread's four
      // (key, position) pairs live in registers from here on.
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t p = 4u * t + uint32_t(e);
        const float c = a == 0 ? s.x[idx[e]] : (a == 1 ? s.y[idx[e]] : s.z[idx[e]]);
        kk[e] = p < cnt ? orderable(c) : 0xFFFFFFFFu;
        pp[e] = idx[e];
      }
      // (c) bitonic sort of (key, position) inside every sub-segment, ascending.  Element i = 4t + e meets i ^ j:
      // inside the thread for j < 4, across lanes (shuffle) above that (nsub <= 256: j <= 128, 32 lanes away).
      for (uint32_t k = 2; k <= nsub; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
          uint32_t ok[4], op[4];
          if (j >= 4u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              ok[e] = __shfl_xor(kk[e], int(j >> 2));
              op[e] = __shfl_xor(pp[e], int(j >> 2));
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              ok[e] = j == 1u ? kk[e ^ 1] : kk[e ^ 2];
              op[e] = j == 1u ? pp[e ^ 1] : pp[e ^ 2];
            }
          }
          // the lower slot of an ascending pair keeps the smaller (key, position); for j, k >= 4 the direction is the
          // same for the thread's four elements (i = 4t + e)
          const uint32_t i0 = 4u * t;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t i = i0 + uint32_t(e);
            const bool lower = (i & j) == 0u, up = ((i & (nsub - 1u)) & k) == 0u;
            const unsigned long long own = ((unsigned long long)kk[e] << 32) | pp[e];
            const unsigned long long oth = ((unsigned long long)ok[e] << 32) | op[e];
            const bool keep_own = (lower == up) == (own < oth);
            kk[e] = keep_own ? kk[e] : ok[e];
            pp[e] = keep_own ? pp[e] : op[e];
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // the wave's readers of perm (this level's idx[], the payload fetch) are done
#pragma unroll
    for (int e = 0; e < 4; ++e) s.perm[4u * t + uint32_t(e)] = uint16_t(pp[e]);
    __builtin_amdgcn_wave_barrier();  // ... and the next level reads what the wave just wrote (LDS is in order per wave)
    KDB_LAP(5);   // binary levels: keys + bitonic sort + permutation
    if (nsub == 32u) break;  // (guards the unsigned loop condition when bottom_nsub is 32)
  }
  __syncthreads();  // the binary levels ran wave by wave: the output pass below reads every wave's slice
  // the last level: `out` is the caller's array, and the position of every original index goes with it
  for (uint32_t p = t; p < cnt; p += KDB_THREADS) {
    const float4 v = in[base + s.perm[p]];
    out[base + p] = v;
    if (rank) rank[__float_as_uint(v.w)] = base + p;
  }
#ifdef PCLHIP_KDB_TICKS
  __syncthreads();
  KDB_LAP(6);   // output gather + rank scatter
  if (threadIdx.x == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_kdb_ticks[i], kdb_acc[i]);
#endif
}
#ifdef PCLHIP_KDB_TICKS
}  // namespace
}  // namespace pclhip
extern "C" __attribute__((visibility("default"))) void pclhip_debug_kdb_ticks(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(pclhip::g_kdb_ticks), 8 * sizeof(unsigned long long));
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(pclhip::g_kdb_ticks), z, sizeof z);
  }
}
namespace pclhip {
namespace {
#endif

// ---- top kd rounds by selection + partition ----------------------------------------------------------
// A round only has to cut every segment into four slabs at its quartile ORDER STATISTICS; the order inside a
// slab is irrelevant (the next round re-orders it along another axis).  Instead of radix-sorting the whole
// cloud by (segment, coordinate) -- 5-7 passes over 12-byte pairs plus a gather -- a round here is
//   keys     k = orderable(coordinate along the segment's widest axis) - orderable(its minimum): an exact,
//            order-preserving unsigned key whose range [0, W] is known per segment (nbits = bit length of W);
//   select   the keys at the three quartile positions by radix selection, up to three digit passes of
//            <= 11 bits from the top of nbits: per-block LDS histograms merged into one global histogram per
//            segment (and per splitter after the first pass), one small scan per pass;
//   classify every point into one of seven classes (< s1, = s1, between, = s2, between, = s3, > s3),
//            count the classes per 4096-point block, prefix the counts over the blocks of each segment;
//   scatter  every point to (segment base + class base + rank inside its class): the segment ends up
//            arranged by class, so the positional slab boundaries (which fall inside the "= s_k" classes)
//            separate smaller keys from larger ones exactly -- ties at a splitter are split by count.
// About 0.85 GB of traffic per round at 10M points instead of ~3 GB, and no rocprim on this path.
constexpr int KP_BLOCK = 4096;     // points per workgroup (segments are multiples of it)
constexpr int KP_THREADS = 256;
constexpr int KP_ROWS = KP_BLOCK / KP_THREADS;  // 16
constexpr int KP_BINS = 2048;

struct SegParam {   // per segment and round
  uint32_t klo;     // orderable(minimum coordinate along the axis)
  uint32_t nbits;   // bit length of orderable(maximum) - klo
  uint32_t axis;
  uint32_t cnt;     // points in the segment (the last one may be partial)
};
struct SelState {   // per segment and splitter
  uint32_t val;     // key bits decided so far (after the last pass: the splitter)
  uint32_t rank;    // remaining rank inside the bin chosen so far
  uint32_t has;     // 0: the quartile position lies beyond the segment's points (splitter = +inf)
  uint32_t pad;
};

__device__ __forceinline__ void kp_digits(uint32_t nbits, int pass, uint32_t& shift, uint32_t& width, uint32_t& up_shift) {
  const uint32_t w1 = nbits < 11u ? nbits : 11u, sh1 = nbits - w1;
  const uint32_t w2 = sh1 < 11u ? sh1 : 11u, sh2 = sh1 - w2;
  if (pass == 1) { shift = sh1; width = w1; up_shift = 32u; }        // up_shift: bits that must match the prefix
  else if (pass == 2) { shift = sh2; width = w2; up_shift = sh1; }
  else { shift = 0u; width = sh2; up_shift = sh2; }
}

// one workgroup per segment: reduce its boxes, pick the widest axis, publish the key range.
// from_parent == 0: the segment's own chunk boxes (kd_load_box_kernel / kd_chunk_box_kernel), chunks_per_seg of them;
// from_parent != 0: the boxes the previous round's scatter left per (block of the parent segment, child slab):
//                   boxes[(block) * 4 + child], chunks_per_seg = blocks of the PARENT segment
__global__ __launch_bounds__(256) void kp_param_kernel(const Box* __restrict__ boxes, uint32_t nchunks,
                                                       uint32_t chunks_per_seg, uint32_t nseg, uint32_t n, uint64_t seg_size,
                                                       SegParam* __restrict__ sp, SelState* __restrict__ sel, int from_parent) {
  const uint32_t sgm = blockIdx.x;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  const uint64_t owner = from_parent ? uint64_t(sgm >> 2) : uint64_t(sgm);
  const uint64_t b = owner * chunks_per_seg;
  uint64_t e = b + chunks_per_seg;
  if (e > nchunks) e = nchunks;
  for (uint64_t i = b + threadIdx.x; i < e; i += 256) {
    const Box bx = from_parent ? boxes[i * 4 + (sgm & 3u)] : boxes[i];
    lo[0] = fminf(lo[0], bx.lo.x); lo[1] = fminf(lo[1], bx.lo.y); lo[2] = fminf(lo[2], bx.lo.z);
    hi[0] = fmaxf(hi[0], bx.hi.x); hi[1] = fmaxf(hi[1], bx.hi.y); hi[2] = fmaxf(hi[2], bx.hi.z);
  }
  __shared__ float red[4][6];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    lo[d] = wave_min(lo[d]);
    hi[d] = wave_max(hi[d]);
  }
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      red[wave][d] = lo[d];
      red[wave][3 + d] = hi[d];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      lo[d] = fminf(fminf(red[0][d], red[1][d]), fminf(red[2][d], red[3][d]));
      hi[d] = fmaxf(fmaxf(red[0][3 + d], red[1][3 + d]), fmaxf(red[2][3 + d], red[3][3 + d]));
    }
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    uint32_t a = 0;
    float best = ex;
    if (ey > best) { best = ey; a = 1; }
    if (ez > best) { a = 2; }
    const uint32_t klo = orderable(lo[a]), khi = orderable(hi[a]);
    const uint32_t w = khi - klo;
    const uint64_t first = uint64_t(sgm) * seg_size;
    const uint32_t cnt = uint32_t((uint64_t(n) - first) < seg_size ? (uint64_t(n) - first) : seg_size);
    SegParam o;
    o.klo = klo;
    o.nbits = w ? 32u - uint32_t(__builtin_clz(w)) : 0u;
    o.axis = a;
    o.cnt = cnt;
    sp[sgm] = o;
    for (uint32_t k = 0; k < 3; ++k) {
      const uint64_t q = uint64_t(k + 1) * (seg_size / 4);   // first position of slab k + 1
      SelState st;
      st.val = 0;
      st.rank = uint32_t(q < cnt ? q : 0);
      st.has = q < cnt ? 1u : 0u;
      st.pad = 0;
      sel[sgm * 3 + k] = st;
    }
  }
}

// digit histograms of one pass.  PASS 1 also materialises the keys (4 bytes per point for the later passes).
// `copies` (a power of two): the blocks of a segment merge into this many copies of its tables (block % copies), which the
// selection adds up -- with one or four segments every block of the grid would otherwise add to the same few hundred
// populated bins (float keys crowd into the bins of the outer binades), and same-address atomics serialise at the memory
// side: hist<1> 64 us instead of 42 in the first two rounds at 10M points.
// Passes 2 and 3 only count keys that share the digits already decided, normally a few per block: such a block adds them
// to the global tables directly and never touches its LDS tables (zeroing and sweeping 24 KB cost more than the 16 KB
// of keys it reads); blocks with many candidates (skewed or tied keys) take the LDS path.
constexpr uint32_t KP_SPARSE_MAX = 128;
template <int PASS>
__global__ __launch_bounds__(KP_THREADS) void kp_hist_kernel(const float4* __restrict__ pts, uint32_t* __restrict__ keys,
                                                             uint32_t n, uint32_t blocks_per_seg,
                                                             const SegParam* __restrict__ sp, const SelState* __restrict__ sel,
                                                             uint32_t* __restrict__ hist, uint32_t copies) {
  constexpr int NH = PASS == 1 ? 1 : 3;
  __shared__ uint32_t h[NH][KP_BINS];
  __shared__ uint32_t candidates;
  const uint32_t sgm = blockIdx.x / blocks_per_seg;
  const SegParam p = sp[sgm];
  uint32_t shift, width, up;
  kp_digits(p.nbits, PASS, shift, width, up);
  if (PASS > 1 && width == 0u) return;   // nothing left to decide for this segment (wave-uniform)
  uint32_t* g = hist + (size_t(sgm) * copies + (blockIdx.x & (copies - 1u))) * 3 * KP_BINS;
  const uint32_t base = blockIdx.x * uint32_t(KP_BLOCK);
  const uint32_t mask = width >= 32u ? 0xFFFFFFFFu : ((1u << width) - 1u);
  if (PASS == 1) {
    for (int i = threadIdx.x; i < NH * KP_BINS; i += KP_THREADS) (&h[0][0])[i] = 0u;
    __syncthreads();
    // eight points per batch, their loads in flight together (see kp_count_kernel)
#pragma unroll
    for (int e0 = 0; e0 < KP_ROWS; e0 += 8) {
      float4 qq[8];
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const uint32_t i = base + uint32_t(e0 + f) * KP_THREADS + threadIdx.x;
        qq[f] = pts[i < n ? i : n - 1u];
      }
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const uint32_t i = base + uint32_t(e0 + f) * KP_THREADS + threadIdx.x;
        if (i < n) {
          const float4 q = qq[f];
          const float c = p.axis == 0u ? q.x : (p.axis == 1u ? q.y : q.z);
          const uint32_t v = orderable(c) - p.klo;
          keys[i] = v;
          atomicAdd(&h[0][(v >> shift) & mask], 1u);
        }
      }
    }
  } else {
    SelState st[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) st[k] = sel[sgm * 3 + k];
    if (threadIdx.x == 0) candidates = 0u;
    uint32_t v[KP_ROWS];
    uint32_t mine = 0;
#pragma unroll
    for (int e = 0; e < KP_ROWS; ++e) {
      const uint32_t i = base + uint32_t(e) * KP_THREADS + threadIdx.x;
      v[e] = i < n ? keys[i] : 0u;
#pragma unroll
      for (int k = 0; k < 3; ++k) mine += (i < n && st[k].has && (v[e] >> up) == (st[k].val >> up)) ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    __syncthreads();
    if ((threadIdx.x & 63u) == 0u && mine) atomicAdd(&candidates, mine);
    __syncthreads();
    const bool sparse = candidates <= KP_SPARSE_MAX;   // block-uniform
    if (!sparse) {
      for (int i = threadIdx.x; i < NH * KP_BINS; i += KP_THREADS) (&h[0][0])[i] = 0u;
      __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < KP_ROWS; ++e) {
      const uint32_t i = base + uint32_t(e) * KP_THREADS + threadIdx.x;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (i < n && st[k].has && (v[e] >> up) == (st[k].val >> up)) {
          const uint32_t b = (v[e] >> shift) & mask;
          if (sparse) atomicAdd(g + k * KP_BINS + b, 1u);
          else atomicAdd(&h[k][b], 1u);
        }
    }
    if (sparse) return;
  }
  __syncthreads();
  const uint32_t nb = 1u << width;
  for (uint32_t i = threadIdx.x; i < uint32_t(NH) * nb; i += KP_THREADS) {
    const uint32_t k = i / nb, b = i - k * nb;
    const uint32_t c = h[k][b];
    if (c) atomicAdd(g + k * KP_BINS + b, c);
  }
}

// one workgroup per (segment, splitter): find the bin that holds the wanted rank, descend into it
template <int PASS>
__global__ __launch_bounds__(256) void kp_select_kernel(const SegParam* __restrict__ sp, SelState* __restrict__ sel,
                                                        const uint32_t* __restrict__ hist, uint32_t copies) {
  const uint32_t sgm = blockIdx.x / 3, k = blockIdx.x % 3;
  const SegParam p = sp[sgm];
  uint32_t shift, width, up;
  kp_digits(p.nbits, PASS, shift, width, up);
  SelState st = sel[sgm * 3 + k];
  if (!st.has || width == 0u) return;
  constexpr int PER = KP_BINS / 256;  // 8 bins per thread
  uint32_t c[PER], sum = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) c[j] = 0u;
  for (uint32_t cp = 0; cp < copies; ++cp) {
    const uint32_t* H = hist + ((size_t(sgm) * copies + cp) * 3 + (PASS == 1 ? 0u : k)) * KP_BINS;
#pragma unroll
    for (int j = 0; j < PER; ++j) c[j] += H[threadIdx.x * PER + j];
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) sum += c[j];
  __shared__ uint32_t scan[256];
  scan[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {   // inclusive Hillis-Steele scan
    const uint32_t v = threadIdx.x >= uint32_t(o) ? scan[threadIdx.x - o] : 0u;
    __syncthreads();
    scan[threadIdx.x] += v;
    __syncthreads();
  }
  const uint32_t incl = scan[threadIdx.x], excl = incl - sum;
  if (st.rank >= excl && st.rank < incl) {   // exactly one thread
    uint32_t cum = excl;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      if (st.rank >= cum && st.rank < cum + c[j]) {
        st.val |= (threadIdx.x * PER + uint32_t(j)) << shift;
        st.rank -= cum;
        sel[sgm * 3 + k] = st;
      }
      cum += c[j];
    }
  }
}

__device__ __forceinline__ uint32_t kp_class(uint32_t v, const SelState* st) {
  // splitters ascend (ranks do); a missing one is +infinity
  if (!st[0].has || v < st[0].val) return 0u;
  if (v == st[0].val) return 1u;
  if (!st[1].has || v < st[1].val) return 2u;
  if (v == st[1].val) return 3u;
  if (!st[2].has || v < st[2].val) return 4u;
  if (v == st[2].val) return 5u;
  return 6u;
}

// class counts per block: counts[block * 8 + class].  A lane counts its sixteen keys in 16-bit fields of two 64-bit
// words (classes 0-3, classes 4-6; a block holds 4096 keys, so a field never overflows), the fields are summed over the
// wavefront with six exchanges per word -- not sixteen rows of seven ballots, which made this pass over 4 bytes per
// point cost 30 us at 10M points.
__global__ __launch_bounds__(KP_THREADS) void kp_count_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                              uint32_t blocks_per_seg, const SelState* __restrict__ sel,
                                                              uint32_t* __restrict__ counts) {
  __shared__ unsigned long long part[KP_THREADS / WAVE][2];
  const uint32_t sgm = blockIdx.x / blocks_per_seg;
  SelState st[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) st[k] = sel[sgm * 3 + k];
  const uint32_t base = blockIdx.x * uint32_t(KP_BLOCK);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  unsigned long long a = 0ull, b = 0ull;
  // the sixteen keys first, all loads in flight together (a load inside `if (i < n)` is followed by its own wait: sixteen
  // memory round trips one after the other -- what this kernel's 17 us at 10M points were); rows past the end read key n - 1
  uint32_t kv[KP_ROWS];
#pragma unroll
  for (int e = 0; e < KP_ROWS; ++e) {
    const uint32_t i = base + uint32_t(e) * KP_THREADS + threadIdx.x;
    kv[e] = keys[i < n ? i : n - 1u];
  }
#pragma unroll
  for (int e = 0; e < KP_ROWS; ++e) {
    const uint32_t i = base + uint32_t(e) * KP_THREADS + threadIdx.x;
    const uint32_t cl = kp_class(kv[e], st);
    a += (i < n && cl < 4u) ? 1ull << (16u * cl) : 0ull;
    b += (i < n && cl >= 4u) ? 1ull << (16u * (cl - 4u)) : 0ull;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o);
    b += __shfl_xor(b, o);
  }
  if (lane == 0) {
    part[wave][0] = a;
    part[wave][1] = b;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const uint32_t k = threadIdx.x;
    unsigned long long w = 0ull;
#pragma unroll
    for (int v = 0; v < KP_THREADS / WAVE; ++v) w += part[v][k >> 2];
    counts[blockIdx.x * 8 + k] = k < 7u ? uint32_t((w >> (16u * (k & 3u))) & 0xFFFFull) : 0u;
  }
}

// one workgroup per segment: counts -> destination of the first point of every (block, class); the seven classes are
// scanned together (one pass over the counts, one ladder of barriers)
__global__ __launch_bounds__(256) void kp_scan_kernel(const uint32_t* __restrict__ counts, uint32_t nblocks,
                                                      uint32_t blocks_per_seg, uint64_t seg_size,
                                                      uint32_t* __restrict__ offsets) {
  const uint32_t sgm = blockIdx.x;
  const uint32_t b0 = sgm * blocks_per_seg;
  const uint32_t nb = (nblocks - b0) < blocks_per_seg ? (nblocks - b0) : blocks_per_seg;
  const uint32_t per = (nb + 255u) / 256u;
  const uint32_t mine0 = threadIdx.x * per < nb ? threadIdx.x * per : nb;
  const uint32_t mine1 = (mine0 + per) < nb ? (mine0 + per) : nb;
  __shared__ uint32_t scan[7][256];
  const uint4* c4 = reinterpret_cast<const uint4*>(counts);
  uint32_t sum[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
  for (uint32_t b = mine0; b < mine1; ++b) {
    const uint4 lo = c4[size_t(b0 + b) * 2], hi = c4[size_t(b0 + b) * 2 + 1];
    sum[0] += lo.x; sum[1] += lo.y; sum[2] += lo.z; sum[3] += lo.w;
    sum[4] += hi.x; sum[5] += hi.y; sum[6] += hi.z;
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) scan[k][threadIdx.x] = sum[k];
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {   // inclusive Hillis-Steele scans, all classes per step
    uint32_t v[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = threadIdx.x >= uint32_t(o) ? scan[k][threadIdx.x - o] : 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 7; ++k) scan[k][threadIdx.x] += v[k];
    __syncthreads();
  }
  uint32_t run[7];
  uint32_t class_base = uint32_t(uint64_t(sgm) * seg_size);
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    run[k] = class_base + scan[k][threadIdx.x] - sum[k];
    class_base += scan[k][255];
  }
  uint4* o4 = reinterpret_cast<uint4*>(offsets);
  for (uint32_t b = mine0; b < mine1; ++b) {
    const uint4 lo = c4[size_t(b0 + b) * 2], hi = c4[size_t(b0 + b) * 2 + 1];
    o4[size_t(b0 + b) * 2] = make_uint4(run[0], run[1], run[2], run[3]);
    o4[size_t(b0 + b) * 2 + 1] = make_uint4(run[4], run[5], run[6], 0u);
    run[0] += lo.x; run[1] += lo.y; run[2] += lo.z; run[3] += lo.w;
    run[4] += hi.x; run[5] += hi.y; run[6] += hi.z;
  }
}

// BOXES: the next round is another partition round and wants the bounding box of every slab this round creates; the
// points pass through here anyway, so every block leaves the boxes of its points per destination slab
// (slab_box[block * 4 + slab]; kp_param_kernel folds them over the blocks of the segment) instead of a separate pass
// over the cloud (kd_chunk_box_kernel, 45 us per round at 10M points).
template <bool BOXES>
__global__ __launch_bounds__(KP_THREADS) void kp_scatter_kernel(const float4* __restrict__ in, const uint32_t* __restrict__ keys,
                                                                uint32_t n, uint32_t blocks_per_seg,
                                                                const SelState* __restrict__ sel,
                                                                const uint32_t* __restrict__ offsets,
                                                                float4* __restrict__ out, uint32_t slab_shift,
                                                                Box* __restrict__ slab_box) {
  constexpr int WAVES = KP_THREADS / WAVE;       // 4
  constexpr int CELLS = KP_ROWS * WAVES;         // 64 (row, wave) cells in position order
  __shared__ uint32_t cell[CELLS][8];
  __shared__ uint32_t boff[8];
  const uint32_t sgm = blockIdx.x / blocks_per_seg;
  SelState st[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) st[k] = sel[sgm * 3 + k];
  if (threadIdx.x < 8) boff[threadIdx.x] = offsets[blockIdx.x * 8 + threadIdx.x];
  const uint32_t base = blockIdx.x * uint32_t(KP_BLOCK);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  uint32_t cls[KP_ROWS], rnk[KP_ROWS];
  // (as in kp_count_kernel: the keys of all rows are asked for before the first is used)
#pragma unroll
  for (int e = 0; e < KP_ROWS; ++e) {
    const uint32_t i = base + uint32_t(e) * KP_THREADS + threadIdx.x;
    cls[e] = keys[i < n ? i : n - 1u];
  }
#pragma unroll
  for (int e = 0; e < KP_ROWS; ++e) {
    const uint32_t i = base + uint32_t(e) * KP_THREADS + threadIdx.x;
    cls[e] = i < n ? kp_class(cls[e], st) : 7u;
    rnk[e] = 0u;
#pragma unroll
    for (uint32_t k = 0; k < 7; ++k) {
      const unsigned long long b = __builtin_amdgcn_ballot_w64(cls[e] == k);
      if (cls[e] == k) rnk[e] = uint32_t(__builtin_popcountll(b & below));
      if (lane == 0) cell[e * WAVES + int(wave)][k] = uint32_t(__builtin_popcountll(b));
    }
  }
  __syncthreads();
  if (threadIdx.x < WAVE) {   // exclusive prefix over the 64 cells, per class (one wavefront, lane = cell)
#pragma unroll
    for (uint32_t k = 0; k < 7; ++k) {
      const uint32_t own = cell[lane][k];
      uint32_t v = own;
#pragma unroll
      for (int o = 1; o < WAVE; o <<= 1) {
        const uint32_t t = __shfl_up(v, o);
        if (lane >= uint32_t(o)) v += t;
      }
      cell[lane][k] = v - own;
    }
  }
  __syncthreads();
  float lo[4][3], hi[4][3];
  if (BOXES) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        lo[c][d] = FLT_MAX;
        hi[c][d] = -FLT_MAX;
      }
  }
  const uint32_t seg_first = sgm * blocks_per_seg * uint32_t(KP_BLOCK);
  // the points in batches of KP_MOVE rows: every load of a batch is in flight before its first store (load, wait, store per
  // row made the partition a chain of sixteen memory round trips per wavefront: 97 us per round at 10M points)
  constexpr int KP_MOVE = 8;
#pragma unroll
  for (int e0 = 0; e0 < KP_ROWS; e0 += KP_MOVE) {
  float4 qq[KP_MOVE];
#pragma unroll
  for (int f = 0; f < KP_MOVE; ++f) {
    const uint32_t i = base + uint32_t(e0 + f) * KP_THREADS + threadIdx.x;
    qq[f] = in[i < n ? i : n - 1u];
  }
#pragma unroll
  for (int f = 0; f < KP_MOVE; ++f) {
    const int e = e0 + f;
    if (cls[e] < 7u) {
      const uint32_t dst = boff[cls[e]] + cell[e * WAVES + int(wave)][cls[e]] + rnk[e];
      const float4 q = qq[f];
      out[dst] = q;
      if (BOXES) {
        const uint32_t slab = (dst - seg_first) >> slab_shift;
#pragma unroll
        for (uint32_t c = 0; c < 4; ++c) {
          const bool mine = slab == c;
          lo[c][0] = fminf(lo[c][0], mine ? q.x : FLT_MAX); hi[c][0] = fmaxf(hi[c][0], mine ? q.x : -FLT_MAX);
          lo[c][1] = fminf(lo[c][1], mine ? q.y : FLT_MAX); hi[c][1] = fmaxf(hi[c][1], mine ? q.y : -FLT_MAX);
          lo[c][2] = fminf(lo[c][2], mine ? q.z : FLT_MAX); hi[c][2] = fmaxf(hi[c][2], mine ? q.z : -FLT_MAX);
        }
      }
    }
  }
  }
  if (BOXES) {
    __shared__ float red[WAVES][4][6];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const float l = wave_min(lo[c][d]), h = wave_max(hi[c][d]);
        if (lane == 0) {
          red[wave][c][d] = l;
          red[wave][c][3 + d] = h;
        }
      }
    __syncthreads();
    if (threadIdx.x < 4) {
      const uint32_t c = threadIdx.x;
      float l[3], h[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        l[d] = fminf(fminf(red[0][c][d], red[1][c][d]), fminf(red[2][c][d], red[3][c][d]));
        h[d] = fmaxf(fmaxf(red[0][c][3 + d], red[1][c][3 + d]), fmaxf(red[2][c][3 + d], red[3][c][3 + d]));
      }
      Box bx;
      bx.lo = make_float4(l[0], l[1], l[2], 0.0f);
      bx.hi = make_float4(h[0], h[1], h[2], 0.0f);
      slab_box[size_t(blockIdx.x) * 4 + c] = bx;
    }
  }
}

// initial compaction: finite selected records first (stable), non-finite ones after them
// `sc`: per-axis rescale factors of a point representation (1,1,1 normally); an axis with factor 0 does not
// exist in the representation: its coordinate reads as 0 and need not be finite
struct Scale3 {
  float x, y, z;
};
__global__ __launch_bounds__(256) void kd_flag_kernel(const void* pts, size_t stride, const int32_t* sel, uint64_t m,
                                                      uint32_t* __restrict__ flags, Scale3 sc) {
  for (uint64_t i = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x; i < m; i += uint64_t(gridDim.x) * blockDim.x) {
    const uint64_t rec = sel ? uint64_t(sel[i]) : i;
    const float* p = record(pts, stride, rec);
    const bool fin = (sc.x == 0.0f || isfinite(p[0])) && (sc.y == 0.0f || isfinite(p[1])) && (sc.z == 0.0f || isfinite(p[2]));
    flags[i] = fin ? 1u : 0u;
  }
}
// stable two-way partition: finite records keep their order in front, the others keep theirs behind them
// (before[i] = exclusive scan of flags = finite records before i)
__global__ __launch_bounds__(256) void kd_compact_kernel(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ before,
                                                         const int32_t* sel, uint64_t m, uint32_t nf,
                                                         uint32_t* __restrict__ vals) {
  const uint64_t i = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
  if (i >= m) return;
  const uint32_t rec = sel ? uint32_t(sel[i]) : uint32_t(i);
  const uint32_t pos = flags[i] ? before[i] : nf + (uint32_t(i) - before[i]);
  vals[pos] = rec;
}

// vals == nullptr: record j of the selection (or of the cloud) goes to slot j
__global__ __launch_bounds__(256) void kd_load_kernel(const void* pts, size_t stride, const uint32_t* vals, const int32_t* sel,
                                                      uint64_t m, float4* out, int ids_from_w, Scale3 sc, int scaled) {
  const uint64_t j = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
  if (j >= m) return;
  const uint32_t rec = vals ? vals[j] : (sel ? uint32_t(sel[j]) : uint32_t(j));
  const float* p = record(pts, stride, rec);
  float x = p[0], y = p[1], z = p[2];
  if (scaled) {  // PointRepresentation::vectorize: coordinate * alpha (common/include/pcl/point_representation.h:150-170)
    x = sc.x == 0.0f ? 0.0f : __fmul_rn(x, sc.x);
    y = sc.y == 0.0f ? 0.0f : __fmul_rn(y, sc.y);
    z = sc.z == 0.0f ? 0.0f : __fmul_rn(z, sc.z);
  }
  // ids_from_w: the records are float4 that already carry the point's id in .w
  out[j] = make_float4(x, y, z, ids_from_w ? p[3] : __uint_as_float(rec));
}

// The common case in one pass: record j of the selection (or of the cloud) -> slot j, the number of finite records, and
// the bounding box of every block of KP_BLOCK slots (the chunk boxes of round 0 and of the first partition round).  If a
// record turns out not to be finite the count says so and the caller takes the compaction path; boxes and slots of this
// pass are then overwritten.
__global__ __launch_bounds__(KP_THREADS) void kd_load_box_kernel(const void* pts, size_t stride, const int32_t* sel, uint64_t m,
                                                                 float4* __restrict__ out, int ids_from_w, Scale3 sc, int scaled,
                                                                 unsigned int* n_finite, Box* __restrict__ chunk_box) {
  const uint64_t base = uint64_t(blockIdx.x) * KP_BLOCK;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  unsigned int mine = 0;
  // eight records per batch, their loads in flight together (see kp_count_kernel); rows past the end read record m - 1
  constexpr int LB = 8;
  for (int e0 = 0; e0 < KP_ROWS; e0 += LB) {
    uint32_t recs[LB];
    float xs[LB], ys[LB], zs[LB], ws[LB];
#pragma unroll
    for (int f = 0; f < LB; ++f) {
      const uint64_t j = base + uint64_t(e0 + f) * KP_THREADS + threadIdx.x;
      const uint64_t jc = j < m ? j : m - 1;
      recs[f] = sel ? uint32_t(sel[jc]) : uint32_t(jc);
    }
#pragma unroll
    for (int f = 0; f < LB; ++f) {
      const float* p = record(pts, stride, recs[f]);
      xs[f] = p[0]; ys[f] = p[1]; zs[f] = p[2];
    }
    if (ids_from_w) {   // (one branch around the batch, not one per record: a branch ends the run of loads)
#pragma unroll
      for (int f = 0; f < LB; ++f) ws[f] = record(pts, stride, recs[f])[3];
    } else {
#pragma unroll
      for (int f = 0; f < LB; ++f) ws[f] = __uint_as_float(recs[f]);
    }
#pragma unroll
    for (int f = 0; f < LB; ++f) {
      const uint64_t j = base + uint64_t(e0 + f) * KP_THREADS + threadIdx.x;
      if (j < m) {
        float x = xs[f], y = ys[f], z = zs[f];
        const bool fin = (sc.x == 0.0f || isfinite(x)) && (sc.y == 0.0f || isfinite(y)) && (sc.z == 0.0f || isfinite(z));
        mine += fin ? 1u : 0u;
        if (scaled) {  // as kd_load_kernel
          x = sc.x == 0.0f ? 0.0f : __fmul_rn(x, sc.x);
          y = sc.y == 0.0f ? 0.0f : __fmul_rn(y, sc.y);
          z = sc.z == 0.0f ? 0.0f : __fmul_rn(z, sc.z);
        }
        out[j] = make_float4(x, y, z, ws[f]);
        lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
        hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
      }
    }
  }
  __shared__ float red[KP_THREADS / WAVE][6];
  __shared__ unsigned int blk;
  if (threadIdx.x == 0) blk = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    lo[d] = wave_min(lo[d]);
    hi[d] = wave_max(hi[d]);
  }
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      red[wave][d] = lo[d];
      red[wave][3 + d] = hi[d];
    }
  }
  if (mine) atomicAdd(&blk, mine);
  __syncthreads();
  if (threadIdx.x == 0) {
    Box bx;
    bx.lo = make_float4(fminf(fminf(red[0][0], red[1][0]), fminf(red[2][0], red[3][0])),
                        fminf(fminf(red[0][1], red[1][1]), fminf(red[2][1], red[3][1])),
                        fminf(fminf(red[0][2], red[1][2]), fminf(red[2][2], red[3][2])), 0.0f);
    bx.hi = make_float4(fmaxf(fmaxf(red[0][3], red[1][3]), fmaxf(red[2][3], red[3][3])),
                        fmaxf(fmaxf(red[0][4], red[1][4]), fmaxf(red[2][4], red[3][4])),
                        fmaxf(fmaxf(red[0][5], red[1][5]), fmaxf(red[2][5], red[3][5])), 0.0f);
    chunk_box[blockIdx.x] = bx;
    if (blk) atomicAdd(n_finite, blk);
  }
}

// what the host waits for after the load pass, written straight into pinned host memory (no copy commands between the
// load and the first round: the stream's idle time there is host latency)
struct LoadResult {
  Box box;
  unsigned int n_finite;
  unsigned int pad[7];
};

// one workgroup: the box of all chunk boxes (the bounding box the caller gets) and the finite count, for the host
__global__ __launch_bounds__(256) void kd_fold_box_kernel(const Box* __restrict__ boxes, uint32_t count,
                                                          const unsigned int* __restrict__ n_finite,
                                                          LoadResult* __restrict__ out) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (uint32_t i = threadIdx.x; i < count; i += 256) {
    const Box bx = boxes[i];
    lo[0] = fminf(lo[0], bx.lo.x); lo[1] = fminf(lo[1], bx.lo.y); lo[2] = fminf(lo[2], bx.lo.z);
    hi[0] = fmaxf(hi[0], bx.hi.x); hi[1] = fmaxf(hi[1], bx.hi.y); hi[2] = fmaxf(hi[2], bx.hi.z);
  }
  __shared__ float red[4][6];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    lo[d] = wave_min(lo[d]);
    hi[d] = wave_max(hi[d]);
  }
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      red[wave][d] = lo[d];
      red[wave][3 + d] = hi[d];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Box bx;
    bx.lo = make_float4(fminf(fminf(red[0][0], red[1][0]), fminf(red[2][0], red[3][0])),
                        fminf(fminf(red[0][1], red[1][1]), fminf(red[2][1], red[3][1])),
                        fminf(fminf(red[0][2], red[1][2]), fminf(red[2][2], red[3][2])), 0.0f);
    bx.hi = make_float4(fmaxf(fmaxf(red[0][3], red[1][3]), fmaxf(red[2][3], red[3][3])),
                        fmaxf(fmaxf(red[0][4], red[1][4]), fmaxf(red[2][4], red[3][4])),
                        fmaxf(fmaxf(red[0][5], red[1][5]), fmaxf(red[2][5], red[3][5])), 0.0f);
    out->box = bx;
    out->n_finite = *n_finite;
  }
}

__global__ __launch_bounds__(256) void kd_finish_kernel(const float4* __restrict__ in, uint64_t live, uint32_t nf,
                                                        float4* __restrict__ out, uint32_t out_cap,
                                                        uint32_t* __restrict__ rank, uint32_t first) {
  const uint64_t j = uint64_t(first) + blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
  if (j >= out_cap) return;
  if (j < live) {
    const float4 p = in[j];
    out[j] = p;
    if (rank && j < nf) rank[__float_as_uint(p.w)] = uint32_t(j);
  } else {
    out[j] = make_float4(FLT_MAX, FLT_MAX, FLT_MAX, __uint_as_float(NO_INDEX));
  }
}

}  // namespace

pclhip_status kd_order(pclhip_ctx* ctx, const void* dev_points, size_t stride, uint64_t n_records,
                       const int32_t* dev_sel, uint64_t n_sel, float4* out_sorted, uint32_t out_capacity,
                       uint32_t* out_n_finite, float lo[3], float hi[3], bool keep_nonfinite_at_end,
                       uint32_t* rank_or_null, bool ids_from_w, const float* scale) {
  hipStream_t s = ctx->stream;
  const Scale3 sc = {scale ? scale[0] : 1.0f, scale ? scale[1] : 1.0f, scale ? scale[2] : 1.0f};
  const int scaled = scale != nullptr ? 1 : 0;
  const uint64_t m = dev_sel ? n_sel : n_records;
  for (int d = 0; d < 3; ++d) lo[d] = hi[d] = 0;
  *out_n_finite = 0;
  if (rank_or_null && n_records) PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(rank_or_null, 0xFF, n_records * sizeof(uint32_t), s));
  if (m == 0) {
    if (out_capacity)
      hipLaunchKernelGGL(kd_finish_kernel, dim3((out_capacity + 255) / 256), dim3(256), 0, s, (const float4*)nullptr,
                         uint64_t(0), 0u, out_sorted, out_capacity, (uint32_t*)nullptr, 0u);
    PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
    return PCLHIP_OK;
  }
  // scratch of the stable compaction of non-finite records: the partials of one scan (device_scan.hpp)
  const size_t temp_bytes = size_t((m + SC_BLOCK - 1) / SC_BLOCK) * sizeof(uint2);
  auto align = [](size_t x) { return (x + 255) & ~size_t(255); };
  const uint32_t max_chunks = uint32_t((m + 63) / 64);
  const size_t off_k0 = 0;
  const size_t off_k1 = off_k0 + align(m * sizeof(uint64_t));
  // second key buffer of the sorting variants; the partition rounds keep their three digit histograms per segment
  // here (segments hold >= 16384 points)
  const size_t hist_room = std::max<size_t>(size_t(m) / 16384 + 1, 64) * 3 * 3 * KP_BINS * sizeof(uint32_t);
  const size_t off_v1 = off_k1 + align(std::max<size_t>(m * sizeof(uint64_t), hist_room));
  const size_t off_pa = off_v1 + align(m * sizeof(uint32_t));
  const size_t off_pb = off_pa + align(m * sizeof(float4));
  const size_t off_cb = off_pb + align(m * sizeof(float4));
  const size_t off_cn = off_cb + align(size_t(max_chunks) * sizeof(Box));
  const size_t off_tmp = off_cn + align(sizeof(unsigned int));
  const size_t total = off_tmp + align(temp_bytes);
  pclhip_status st = ensure_scratch(ctx, total);
  if (st != PCLHIP_OK) return st;
  char* base = static_cast<char*>(ctx->scratch);
  uint64_t* k0 = reinterpret_cast<uint64_t*>(base + off_k0);
  uint64_t* k1 = reinterpret_cast<uint64_t*>(base + off_k1);
  uint32_t* v1 = reinterpret_cast<uint32_t*>(base + off_v1);
  float4* pa = reinterpret_cast<float4*>(base + off_pa);
  float4* pb = reinterpret_cast<float4*>(base + off_pb);
  Box* cb = reinterpret_cast<Box*>(base + off_cb);
  unsigned int* cn = reinterpret_cast<unsigned int*>(base + off_cn);
  void* tmp = base + off_tmp;

  // --- load: record j -> slot j, counting the finite ones and boxing every KP_BLOCK slots on the way; only a cloud
  //     with non-finite records pays the stable compaction (finite records first, the others behind them) ---
  uint32_t* f0 = reinterpret_cast<uint32_t*>(k0);
  uint32_t* f1 = reinterpret_cast<uint32_t*>(k1);
  unsigned int hn = 0;
  const uint32_t load_chunks = uint32_t((m + KP_BLOCK - 1) / KP_BLOCK);
  std::vector<Box> hb(1);
  PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(cn, 0, sizeof(unsigned int), s));
  hipLaunchKernelGGL(kd_load_box_kernel, dim3(load_chunks), dim3(KP_THREADS), 0, s, dev_points, stride, dev_sel, m, pa,
                     ids_from_w ? 1 : 0, sc, scaled, cn, cb);
  {
    LoadResult* res = nullptr;
    PCLHIP_CHECK_HIP(ctx, pinned_malloc(ctx, &res, sizeof(LoadResult)));
    hipLaunchKernelGGL(kd_fold_box_kernel, dim3(1), dim3(256), 0, s, cb, load_chunks, cn, res);
    const hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess) {
      hn = res->n_finite;
      hb[0] = res->box;
    }
    pinned_free(ctx, res, sizeof(LoadResult));
    PCLHIP_CHECK_HIP(ctx, e);
  }
  bool boxes_of_load = uint64_t(hn) == m;   // cb holds the boxes of pa's KP_BLOCK chunks, hb[0] the box of them all
  if (!boxes_of_load) {  // stable compaction: flags -> exclusive scan (device_scan.hpp) -> scatter of the record numbers
    uint32_t* flags = f0;
    uint32_t* before = f1;
    uint2* sc_part = reinterpret_cast<uint2*>(tmp);
    uint32_t* sc_tot = reinterpret_cast<uint32_t*>(cb);   // 16 bytes of a region that is not in use yet
    const unsigned grid = unsigned(std::min<uint64_t>((m + 255) / 256, uint64_t(ctx->num_cus) * 16));
    hipLaunchKernelGGL(kd_flag_kernel, dim3(grid), dim3(256), 0, s, dev_points, stride, dev_sel, m, flags, sc);
    launch_scan_u32(s, flags, m, sc_part, sc_tot, before);
    hipLaunchKernelGGL(kd_compact_kernel, dim3(unsigned((m + 255) / 256)), dim3(256), 0, s, flags, before, dev_sel, m, hn, v1);
    hipLaunchKernelGGL(kd_load_kernel, dim3(unsigned((m + 255) / 256)), dim3(256), 0, s, dev_points, stride, v1,
                       (const int32_t*)nullptr, m, pa, ids_from_w ? 1 : 0, sc, scaled);
  }
  const uint32_t nf = hn;
  *out_n_finite = nf;

  // --- kd rounds over the finite prefix ---
  float4* cur = pa;
  float4* nxt = pb;
  bool placed = false;   // kd_block_kernel has written out_sorted[0, nf) and the ranks
  if (nf > 0) {
    const uint64_t nleaf = (uint64_t(nf) + LEAF - 1) / LEAF;
    int R = 0;
    uint64_t cap = 1;
    while (cap < nleaf) {
      cap *= 4;
      ++R;
    }
    auto fold = [&](const std::vector<Box>& boxes) {
      float l[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, h[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
      for (const Box& b : boxes) {
        l[0] = std::fmin(l[0], b.lo.x); l[1] = std::fmin(l[1], b.lo.y); l[2] = std::fmin(l[2], b.lo.z);
        h[0] = std::fmax(h[0], b.hi.x); h[1] = std::fmax(h[1], b.hi.y); h[2] = std::fmax(h[2], b.hi.z);
      }
      for (int d = 0; d < 3; ++d) {
        lo[d] = l[d];
        hi[d] = h[d];
      }
    };
    bool slab_boxes = false;            // cb holds the previous scatter's boxes per (block, slab)
    uint32_t parent_blocks = 0;         // ... of segments of this many blocks
    // round 0 (no sort) only measures the bounding box of everything for the caller
    for (int r = 0; r <= R; ++r) {
      // segment being split this round (r = 0: one segment covering all points, bbox only)
      uint64_t seg_size = uint64_t(LEAF);
      for (int j = 0; j < R - r + 1; ++j) seg_size *= 4;
      if (r == 0) seg_size = uint64_t(LEAF) * cap * 4;
      uint32_t chunk = seg_size < uint64_t(KD_CHUNK_MAX) ? uint32_t(seg_size) : uint32_t(KD_CHUNK_MAX);
      const uint32_t nchunks = uint32_t((uint64_t(nf) + chunk - 1) / chunk);
      const uint32_t chunks_per_seg = uint32_t(seg_size / chunk);
      const uint32_t nseg = uint32_t((uint64_t(nf) + seg_size - 1) / seg_size);
      if (r > 0 && seg_size <= uint64_t(KDB_N)) {  // the remaining rounds fit one workgroup's LDS: the last launch
        hipLaunchKernelGGL(kd_block_kernel, dim3(unsigned((uint64_t(nf) + KDB_N - 1) / KDB_N)), dim3(KDB_THREADS), 0, s, cur, nf,
                           out_sorted, uint32_t(seg_size), 32u, rank_or_null);
        placed = true;
        break;
      }
      if (r == 0) {
        if (boxes_of_load) {
          fold(hb);
          continue;
        }
        hipLaunchKernelGGL(kd_chunk_box_kernel, dim3(unsigned((uint64_t(nchunks) * WAVE + 255) / 256)), dim3(256), 0, s, cur,
                           nf, chunk, nchunks, cb);
        hb.resize(nchunks);   // nseg == 1: reduce on the host (tiny)
        PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(hb.data(), cb, size_t(nchunks) * sizeof(Box), hipMemcpyDeviceToHost, s));
        PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
        fold(hb);
        // the same boxes serve the first partition round when its chunks are these
        boxes_of_load = chunk == uint32_t(KP_BLOCK);
        continue;
      }
      if (chunk == uint32_t(KP_BLOCK) && seg_size % KP_BLOCK == 0) {
        // selection + partition round (see kp_* above): blocks of 4096 points, each inside one segment
        const uint32_t nblocks = nchunks, blocks_per_seg = chunks_per_seg;
        uint32_t* keys = reinterpret_cast<uint32_t*>(k0);
        uint32_t* hist = reinterpret_cast<uint32_t*>(k1);
        SegParam* sp = reinterpret_cast<SegParam*>(v1);
        SelState* sel = reinterpret_cast<SelState*>(sp + nseg);
        // copies of a segment's tables (see kp_hist_kernel): nseg * copies <= 64
        const uint32_t copies = nseg <= 4u ? 16u : (nseg <= 16u ? 4u : 1u);
        const size_t hist_bytes = size_t(nseg) * copies * 3 * KP_BINS * sizeof(uint32_t);
        // class counts per block and the destinations scanned from them: 2 x 8 words per block in the key buffer's
        // upper half (keys are 4 bytes per point, the buffer holds 8)
        uint32_t* counts = reinterpret_cast<uint32_t*>(k0) + ((size_t(m) + 3) & ~size_t(3));   // 16-byte aligned
        uint32_t* offsets = counts + size_t(nblocks) * 8;
        if (slab_boxes) {
          hipLaunchKernelGGL(kp_param_kernel, dim3(nseg), dim3(256), 0, s, cb, nblocks, parent_blocks, nseg, nf, seg_size, sp, sel,
                             1);
        } else {
          if (!(r == 1 && boxes_of_load))
            hipLaunchKernelGGL(kd_chunk_box_kernel, dim3(unsigned((uint64_t(nchunks) * WAVE + 255) / 256)), dim3(256), 0, s, cur,
                               nf, chunk, nchunks, cb);
          hipLaunchKernelGGL(kp_param_kernel, dim3(nseg), dim3(256), 0, s, cb, nchunks, chunks_per_seg, nseg, nf, seg_size, sp,
                             sel, 0);
        }
        PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(hist, 0, 3 * hist_bytes, s));   // one table per digit pass
        uint32_t* hist2 = hist + hist_bytes / sizeof(uint32_t);
        uint32_t* hist3 = hist2 + hist_bytes / sizeof(uint32_t);
        hipLaunchKernelGGL(kp_hist_kernel<1>, dim3(nblocks), dim3(KP_THREADS), 0, s, cur, keys, nf, blocks_per_seg, sp, sel, hist,
                           copies);
        hipLaunchKernelGGL(kp_select_kernel<1>, dim3(nseg * 3), dim3(256), 0, s, sp, sel, hist, copies);
        hipLaunchKernelGGL(kp_hist_kernel<2>, dim3(nblocks), dim3(KP_THREADS), 0, s, cur, keys, nf, blocks_per_seg, sp, sel, hist2,
                           copies);
        hipLaunchKernelGGL(kp_select_kernel<2>, dim3(nseg * 3), dim3(256), 0, s, sp, sel, hist2, copies);
        hipLaunchKernelGGL(kp_hist_kernel<3>, dim3(nblocks), dim3(KP_THREADS), 0, s, cur, keys, nf, blocks_per_seg, sp, sel, hist3,
                           copies);
        hipLaunchKernelGGL(kp_select_kernel<3>, dim3(nseg * 3), dim3(256), 0, s, sp, sel, hist3, copies);
        hipLaunchKernelGGL(kp_count_kernel, dim3(nblocks), dim3(KP_THREADS), 0, s, keys, nf, blocks_per_seg, sel, counts);
        hipLaunchKernelGGL(kp_scan_kernel, dim3(nseg), dim3(256), 0, s, counts, nblocks, blocks_per_seg, seg_size, offsets);
        // the slabs this round creates are the next round's segments: box them here unless kd_block_kernel comes next
        const bool next_is_partition = seg_size / 4 > uint64_t(KDB_N);
        uint32_t slab_shift = 0;
        while ((uint64_t(1) << slab_shift) < seg_size / 4) ++slab_shift;
        if (next_is_partition)
          hipLaunchKernelGGL(kp_scatter_kernel<true>, dim3(nblocks), dim3(KP_THREADS), 0, s, cur, keys, nf, blocks_per_seg, sel,
                             offsets, nxt, slab_shift, cb);
        else
          hipLaunchKernelGGL(kp_scatter_kernel<false>, dim3(nblocks), dim3(KP_THREADS), 0, s, cur, keys, nf, blocks_per_seg, sel,
                             offsets, nxt, slab_shift, cb);

        slab_boxes = next_is_partition;
        parent_blocks = blocks_per_seg;
        if (keep_nonfinite_at_end && m > nf)
          PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(nxt + nf, cur + nf, (m - nf) * sizeof(float4), hipMemcpyDeviceToDevice, s));
        float4* t = cur;
        cur = nxt;
        nxt = t;
        continue;
      }
      // every segment above KDB_N points is a multiple of KP_BLOCK (16 * 4^j >= 4096) and everything below goes
      // through kd_block_kernel: no other round shape exists
      set_error(ctx, "kd order: unexpected round shape");
      return PCLHIP_ERR_STATE;
    }
  }
  // what kd_block_kernel did not place: the non-finite records behind the finite ones (if kept) and the sentinels of
  // the capacity -- or everything, for a cloud of at most one leaf, which has no round at all
  const uint64_t live = keep_nonfinite_at_end ? m : nf;
  const uint32_t first = placed ? nf : 0u;
  if (out_capacity > first)
    hipLaunchKernelGGL(kd_finish_kernel, dim3((out_capacity - first + 255) / 256), dim3(256), 0, s, cur, live, nf, out_sorted,
                       out_capacity, rank_or_null, first);
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
  return PCLHIP_OK;
}

pclhip_status spatial_order(pclhip_ctx* ctx, const void* dev_points, size_t stride, uint64_t n_records,
                            const int32_t* dev_sel, uint64_t n_sel, float4* out_sorted, uint32_t out_capacity,
                            uint32_t* out_n_finite, float lo[3], float hi[3], bool keep_nonfinite_at_end,
                            uint32_t* rank_or_null, const float* scale) {
  return kd_order(ctx, dev_points, stride, n_records, dev_sel, n_sel, out_sorted, out_capacity, out_n_finite, lo, hi,
                  keep_nonfinite_at_end, rank_or_null, false, scale);
}

pclhip_status build_index_from_float4(pclhip_ctx* ctx, const float4* dev_pts_with_ids, uint32_t n, pclhip_index** out) {
  *out = nullptr;
  pclhip_index* ix = new pclhip_index();
  ix->ctx = ctx;
  ix->n_orig = n;
  const uint32_t cap = ((n + LEAF - 1) / LEAF) * LEAF + LEAF;
  if (dev_malloc(ctx, &ix->pts, size_t(cap) * sizeof(float4)) != hipSuccess) {
    delete ix;
    set_error(ctx, "hipMalloc failed for the reciprocal index");
    return PCLHIP_ERR_HIP;
  }
  uint32_t nf = 0;
  pclhip_status st = kd_order(ctx, dev_pts_with_ids, sizeof(float4), n, nullptr, 0, ix->pts, cap, &nf, ix->bbox_lo,
                              ix->bbox_hi, false, nullptr, true, nullptr);
  if (st == PCLHIP_OK) {
    ix->n = nf;
    ix->n_pad = ((nf + LEAF - 1) / LEAF) * LEAF;
    if (ix->n_pad == 0) ix->n_pad = LEAF;
    st = build_boxes(ix);
  }
  if (st != PCLHIP_OK) {
    pclhip_index_destroy(ix);
    return st;
  }
  *out = ix;
  return PCLHIP_OK;
}

// An index over points that ARE in kd order already -- the working copy of a registration's source cloud, ordered by
// spatial_order() when the source was set: nothing is sorted or copied, the index BORROWS the array (w = original index,
// the finite points in front) and only carries boxes and the leaf blocks of it.  refit_boxes() follows the cloud.
pclhip_status build_index_over(pclhip_ctx* ctx, float4* pts_in_kd_order, uint32_t n_finite, uint32_t n_orig, pclhip_index** out) {
  *out = nullptr;
  pclhip_index* ix = new pclhip_index();
  ix->ctx = ctx;
  ix->n_orig = n_orig;
  ix->pts = pts_in_kd_order;
  ix->pts_borrowed = true;
  ix->n = n_finite;
  ix->n_pad = ((n_finite + LEAF - 1) / LEAF) * LEAF;
  if (ix->n_pad == 0) ix->n_pad = LEAF;
  const pclhip_status st = build_boxes(ix, false);
  if (st != PCLHIP_OK) {
    pclhip_index_destroy(ix);
    return st;
  }
  *out = ix;
  return PCLHIP_OK;
}

pclhip_status build_boxes(pclhip_index* ix, bool with_discs) {
  pclhip_ctx* ctx = ix->ctx;
  hipStream_t s = ctx->stream;
  constexpr int DIAG_NB = 512;  // 64 until late in round 6: 16K lanes walking 625K leaves took 27 us at 10M points (now 10)
  double* diag_part = nullptr;   // pinned; released at the final wait
  uint32_t diag_leaves = 0;
  struct PinnedGuard {
    pclhip_ctx* ctx;
    double** p;
    size_t bytes;
    ~PinnedGuard() { if (*p) pinned_free(ctx, *p, bytes); }
  } diag_guard{ctx, &diag_part, 3 * DIAG_NB * sizeof(double)};
  if (ix->qbox) ix->box[1] = nullptr;  // box[1] was the front of qbox
  for (int l = 0; l < MAX_LEVELS; ++l) {
    if (ix->box[l]) (void)dev_free(ctx, ix->box[l]);
    ix->box[l] = nullptr;
    ix->count[l] = 0;
  }
  if (ix->qbox) (void)dev_free(ctx, ix->qbox);
  if (ix->qcell) (void)dev_free(ctx, ix->qcell);
  ix->qbox = ix->qcell = nullptr;
  ix->qtop = 0;
  ix->count[0] = ix->n;
  uint32_t c = (ix->n + LEAF - 1) / LEAF;
  if (c == 0) c = 1;  // an empty index still has one (empty) leaf so kernels stay uniform
  // the per-lane search structure of a kd index (LaneTree): every quad level in one array, the leaf boxes in front
  const bool lane_tree = with_discs;
  size_t qnodes = 0;
  if (lane_tree) {
    int q = 0;
    for (;; ++q) {
      const uint32_t cq = lane_tree_count(c, q);
      qnodes += cq;
      if (cq == 1) break;
    }
    ix->qtop = q;
  }
  int l = 1;
  for (;;) {
    if (l >= MAX_LEVELS) {
      set_error(ctx, "index too large for MAX_LEVELS");
      return PCLHIP_ERR_INVALID;
    }
    ix->count[l] = c;
    if (l == 1 && lane_tree) {
      PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &ix->qbox, qnodes * sizeof(Box)));
      PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &ix->qcell, qnodes * sizeof(Box)));
      ix->box[1] = ix->qbox;
    } else {
      PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &ix->box[l], size_t(c) * sizeof(Box)));
    }
    if (l == 1) {
      const uint32_t threads = c * LEAF;
      if (ix->soa) (void)dev_free(ctx, ix->soa);
      ix->soa = nullptr;
      PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &ix->soa, size_t(c) * 4 * LEAF * sizeof(float)));
      hipLaunchKernelGGL(leaf_box_soa_kernel, dim3((threads + 255) / 256), dim3(256), 0, s, ix->pts, ix->n, c, ix->box[1],
                         ix->soa);
      if (ix->disc) (void)dev_free(ctx, ix->disc);
      ix->disc = nullptr;
      if (with_discs) {
      PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &ix->disc, size_t(c) * 2 * sizeof(float4)));
      hipLaunchKernelGGL(leaf_disc_kernel, dim3((c + 255) / 256), dim3(256), 0, s, ix->pts, ix->n, c, ix->disc);
      // mean squared leaf diagonal: from what stand-off on the discs pay (traverse.hpp).  The partial sums land in
      // pinned host memory and are read at this function's final wait, so the stream does not idle here.
      if (pinned_malloc(ctx, &diag_part, 3 * DIAG_NB * sizeof(double)) != hipSuccess) {
        set_error(ctx, "hipHostMalloc failed for the leaf statistics");
        return PCLHIP_ERR_HIP;
      }
      diag_leaves = c;
      hipLaunchKernelGGL(leaf_diag_kernel, dim3(DIAG_NB), dim3(256), 0, s, ix->box[1], ix->disc, c, diag_part);
      }  // with_discs
    } else {
      const uint64_t threads = uint64_t(c) * WAVE;
      hipLaunchKernelGGL(node_box_kernel, dim3(unsigned((threads + 255) / 256)), dim3(256), 0, s, ix->box[l - 1],
                         ix->count[l - 1], ix->box[l], c);
    }
    if (c <= uint32_t(FANOUT) && l >= 1) break;
    c = (c + FANOUT - 1) / FANOUT;
    ++l;
  }
  ix->top = l;
  if (lane_tree) {  // quad levels bottom-up (boxes), then top-down (cells): ~2 x 13 small launches at 10M points
    const uint32_t nleaf = ix->count[1];
    size_t off = 0;
    // the large levels one launch each; the levels of at most QUAD_TOP_NODES nodes together in one workgroup
    // (quad_top_kernel: boxes up, cells down) -- 20 launches become 8 at 10M points
    constexpr uint32_t QUAD_TOP_NODES = 4096;
    int q_small = ix->qtop + 1;   // first level built by the fused kernel
    for (int q = 1; q <= ix->qtop; ++q)
      if (lane_tree_count(nleaf, q) <= QUAD_TOP_NODES) {
        q_small = q;
        break;
      }
    for (int q = 1; q < q_small; ++q) {
      const uint32_t cc = lane_tree_count(nleaf, q - 1), cp = lane_tree_count(nleaf, q);
      hipLaunchKernelGGL(quad_box_kernel, dim3((cp + 255) / 256), dim3(256), 0, s, ix->qbox + off, cc, ix->qbox + off + cc, cp);
      off += cc;
    }
    // off = offset of level q_small - 1 now
    if (q_small <= ix->qtop)
      hipLaunchKernelGGL(quad_top_kernel, dim3(1), dim3(1024), 0, s, ix->qbox, ix->qcell, nleaf, q_small, ix->qtop);
    for (int q = q_small - 1; q >= 1; --q) {
      const uint32_t cc = lane_tree_count(nleaf, q - 1), cp = lane_tree_count(nleaf, q);
      const size_t coff = off - cc;
      hipLaunchKernelGGL(quad_cell_kernel, dim3((cp + 255) / 256), dim3(256), 0, s, ix->qcell + off, cp, q == ix->qtop ? 1 : 0,
                         ix->qbox + coff, cc, ix->qcell + coff);
      off = coff;
    }
  }
  // contiguous copy of the top levels (as many whole levels as fit TOPCACHE_BOXES, never the leaves)
  {
    uint32_t total = 0;
    int from = MAX_LEVELS;
    for (int lv = ix->top; lv >= 2; --lv) {
      if (total + ix->count[lv] > uint32_t(TOPCACHE_BOXES)) break;
      total += ix->count[lv];
      from = lv;
    }
    ix->cache_from = from;
    ix->cache_count = total;
    if (!ix->topcache) PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &ix->topcache, size_t(TOPCACHE_BOXES) * sizeof(Box)));
    uint32_t off = 0;
    for (int lv = 0; lv < MAX_LEVELS; ++lv) ix->cache_off[lv] = 0;
    for (int lv = from; lv <= ix->top && from < MAX_LEVELS; ++lv) {
      ix->cache_off[lv] = off;
      PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(ix->topcache + off, ix->box[lv], size_t(ix->count[lv]) * sizeof(Box),
                                           hipMemcpyDeviceToDevice, s));
      off += ix->count[lv];
    }
  }
  LevelInfo h[MAX_LEVELS];
  for (int i = 0; i < MAX_LEVELS; ++i) {
    h[i].box = ix->box[i];
    h[i].count = ix->count[i];
    h[i].pad = 0;
  }
  if (!ix->lv_dev) PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &ix->lv_dev, sizeof h));
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(ix->lv_dev, h, sizeof h, hipMemcpyHostToDevice, s));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));  // h is a stack buffer
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  if (diag_part) {
    double sum = 0.0, hs = 0.0, rs = 0.0;
    for (int i = 0; i < DIAG_NB; ++i) {
      sum += diag_part[i];
      hs += diag_part[DIAG_NB + i];
      rs += diag_part[2 * DIAG_NB + i];
    }
    ix->leaf_diag2 = float(sum / double(diag_leaves));
    ix->disc_thickness = rs > 0.0 ? float(hs / rs) : 1.0f;
  }
  return PCLHIP_OK;
}


// The boxes of an index whose points MOVED (same count, same order): leaf boxes, the SoA copy and the upper levels are
// recomputed from ix->pts; nothing is allocated and nothing waits for the stream.  The order stays a valid grouping
// under a rigid motion, but the leaves are no longer cells of an axis-aligned kd partition: searches of such an index
// must start at the root (no start-level shortcut) and it carries no discs.
pclhip_status refit_boxes(pclhip_index* ix) {
  pclhip_ctx* ctx = ix->ctx;
  hipStream_t s = ctx->stream;
  if (ix->top < 1 || ix->box[1] == nullptr || ix->soa == nullptr) {
    set_error(ctx, "refit of an index that was never built");
    return PCLHIP_ERR_STATE;
  }
  const uint32_t c1 = ix->count[1];
  const uint32_t threads = c1 * LEAF;
  hipLaunchKernelGGL(leaf_box_soa_kernel, dim3((threads + 255) / 256), dim3(256), 0, s, ix->pts, ix->n, c1, ix->box[1],
                     ix->soa);
  for (int l = 2; l <= ix->top; ++l) {
    const uint64_t th = uint64_t(ix->count[l]) * WAVE;
    hipLaunchKernelGGL(node_box_kernel, dim3(unsigned((th + 255) / 256)), dim3(256), 0, s, ix->box[l - 1], ix->count[l - 1],
                       ix->box[l], ix->count[l]);
  }
  for (int lv = ix->cache_from; lv <= ix->top && ix->cache_from < MAX_LEVELS; ++lv)
    PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(ix->topcache + ix->cache_off[lv], ix->box[lv], size_t(ix->count[lv]) * sizeof(Box),
                                         hipMemcpyDeviceToDevice, s));
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  return PCLHIP_OK;
}

void preload_index_build_kernels() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(leaf_box_soa_kernel));
}

}  // namespace pclhip

pclhip::LaneTree pclhip_index::lane_tree() const {
  pclhip::LaneTree t;
  t.qbox = qbox;
  t.qcell = qcell;
  t.nleaf = count[1];
  t.top = qtop;
  return t;
}

pclhip::IndexView pclhip_index::view() const {
  pclhip::IndexView v;
  v.pts = pts;
  v.soa = soa;
  v.nrm = nrm;
  v.disc = disc;
  v.disc_from = 4.0f * leaf_diag2;  // stand-off (squared, in leaf diagonals squared) from which discs replace boxes
  v.lv = lv_dev;
  for (int l = 0; l < pclhip::MAX_LEVELS; ++l) {
    v.box[l] = box[l];
    v.count[l] = count[l];
    v.cache_off[l] = cache_off[l];
  }
  v.topcache = topcache;
  v.cache_from = cache_from;
  v.cache_count = cache_count;
  v.top = top;
  v.n = n;
  v.n_pad = n_pad;
  v.sched_ctr = ctx->sched_ctr;
  v.cell2 = v.cell3 = nullptr;
  if (qcell != nullptr && ctx->opt_cell_start != 0) {
    size_t off = 0;
    for (int q = 0; q <= qtop; ++q) {
      if (q == 3 && q < qtop) v.cell2 = qcell + off;   // (the root's cell is never stored)
      if (q == 6 && q < qtop) v.cell3 = qcell + off;
      off += pclhip::lane_tree_count(count[1], q);
    }
  }
  return v;
}
