// radius.hip -- batched radiusSearch on the same index.
// Replaces pcl::KdTreeFLANN<PointT>::radiusSearch (kdtree/include/pcl/kdtree/impl/kdtree_flann.hpp
// :372-414): all indexed points with squared distance < float(radius*radius) (FLANN's
// RadiusResultSet keeps `dist < radius`), ascending by (distance, index), optionally only the
// max_nn nearest.  Two traversals (count, then fill at exclusive-scan offsets) + one segmented sort of
// (distance, index) keys -- scan and sort hand-written (segsort.hpp).
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>

#include <cfloat>
#include <cmath>

#include "traverse.hpp"
#include "normals_math.hpp"
#include "segsort.hpp"

namespace pclhip {
namespace {

constexpr int BLOCK = 256;
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;

struct RadiusCount {
  static constexpr int QPL = 1;
  float r2;
  uint32_t cnt;
  __device__ __forceinline__ float worst(int) const { return r2; }
  __device__ __forceinline__ void leaf(const float* l, uint32_t, const float* qx, const float* qy, const float* qz) {
    const v2f qx2 = {qx[0], qx[0]}, qy2 = {qy[0], qy[0]}, qz2 = {qz[0], qz[0]};
#pragma unroll
    for (int j = 0; j < LEAF / 2; ++j) {
      const v2f r = pair_dist(l, j, qx2, qy2, qz2);
      cnt += (r.x < r2 ? 1u : 0u) + (r.y < r2 ? 1u : 0u);
    }
  }
};

struct RadiusFill {
  static constexpr int QPL = 1;
  float r2;
  uint64_t* out;  // this query's segment
  uint32_t cnt;
  bool active;
  __device__ __forceinline__ float worst(int) const { return r2; }
  __device__ __forceinline__ void leaf(const float* l, uint32_t, const float* qx, const float* qy, const float* qz) {
    const v2f qx2 = {qx[0], qx[0]}, qy2 = {qy[0], qy[0]}, qz2 = {qz[0], qz[0]};
#pragma unroll
    for (int j = 0; j < LEAF / 2; ++j) {
      const v2f r = pair_dist(l, j, qx2, qy2, qz2);
      if (active && r.x < r2) out[cnt++] = make_key(r.x, __float_as_uint(l[3 * LEAF + 2 * j]));
      if (active && r.y < r2) out[cnt++] = make_key(r.y, __float_as_uint(l[3 * LEAF + 2 * j + 1]));
    }
  }
};

// BYPOS: counts / offsets are indexed by the query's position in `q` (self-queries of the index in kd
// order) instead of by the original index stored in q[i].w; [q_begin, q_end) restricts the launch to a
// chunk of queries whose segments start at offsets[i] - base.
// (box-only LDS layout -- a radius query never bounds leaves by discs --: 36 KB per block, a fourth block per CU)
template <bool FILL, bool BYPOS = false>
__global__ __launch_bounds__(BLOCK, 4) void radius_kernel(IndexView ix, const float4* __restrict__ q, uint32_t nq, float r2,
                                                       uint32_t* __restrict__ counts,
                                                       const unsigned long long* __restrict__ offsets,
                                                       uint64_t* __restrict__ keys, uint32_t q_begin = 0,
                                                       unsigned long long base = 0) {
  __shared__ WaveLdsBoxT<LEAF_BATCH * LEAF_FLOATS * 4> wl_s[WAVES_PER_BLOCK];
  __shared__ Box topbox_s[TOPCACHE_BOXES];
  load_top_cache(ix, topbox_s);
  const int lane = threadIdx.x & (WAVE - 1);
  const uint32_t ngroups = (nq - q_begin + WAVE - 1) / WAVE;
  const GroupSchedule sched(ngroups);
  TraverseStats ts;
  GroupFeed feed(sched, ix.sched_ctr);
  for (uint32_t gl = feed.first(sched); gl != GroupFeed::END; gl = feed.advance()) {
    const uint32_t g = sched.global(gl);
    if (g >= ngroups) break;
    feed.ahead(gl);
    const uint32_t i = q_begin + g * WAVE + lane;
    float4 p = make_float4(0, 0, 0, 0);
    const bool real = i < nq;
    if (real) p = q[i];
    const uint32_t oq = BYPOS ? i : __float_as_uint(p.w);
    const bool vv[1] = {real && isfinite(p.x) && isfinite(p.y) && isfinite(p.z)};
    const float qx[1] = {p.x}, qy[1] = {p.y}, qz[1] = {p.z};
    if constexpr (!FILL) {
      RadiusCount pol;
      pol.r2 = r2;
      pol.cnt = 0;
      traverse(ix, qx, qy, qz, vv, pol, wl_s[threadIdx.x / WAVE], topbox_s, ts);
      if (real) counts[oq] = vv[0] ? pol.cnt : 0u;
    } else {
      RadiusFill pol;
      pol.r2 = r2;
      pol.cnt = 0;
      pol.active = vv[0];
      pol.out = keys + (real ? offsets[oq] - base : 0ull);
      traverse(ix, qx, qy, qz, vv, pol, wl_s[threadIdx.x / WAVE], topbox_s, ts);
    }
  }
}

// sorted full segments -> (possibly truncated) output CSR
__global__ void radius_emit_kernel(const uint64_t* __restrict__ keys, const unsigned long long* __restrict__ full_off,
                                   const unsigned long long* __restrict__ out_off, uint32_t nq,
                                   int32_t* __restrict__ out_idx, float* __restrict__ out_d2) {
  const uint32_t qi = blockIdx.x;
  if (qi >= nq) return;
  const unsigned long long src = full_off[qi], dst = out_off[qi], cnt = out_off[qi + 1] - out_off[qi];
  for (unsigned long long t = threadIdx.x; t < cnt; t += blockDim.x) {
    const uint64_t k = keys[src + t];
    out_idx[dst + t] = int32_t(uint32_t(k));
    out_d2[dst + t] = __uint_as_float(uint32_t(k >> 32));
  }
}

// NormalEstimation with setRadiusSearch (features/include/pcl/features/impl/normal_3d.hpp:48-95 through
// Feature::compute, impl/feature.hpp:140-155): plane fit over ALL neighbours within the radius.  The covariance is
// accumulated INSIDE the radius traversal -- no neighbour lists, no sort: computeMeanAndCovarianceMatrix
// (common/include/pcl/common/impl/centroid.hpp:581-650) shifts by the first neighbour and sums float products in
// ascending-distance order; here the shift is the query itself (the first neighbour of a self-query, coordinate for
// coordinate) and the nine float products are summed in DOUBLE in traversal order, rounded to float once at the end --
// a sum that does not depend on the order to within 2^-53 and sits inside the rounding of the reference's own float
// sum (the plane fit's contract is |n . n_ref| >= 1 - 1e-5).  Fewer than 3 neighbours -> NaN
// (normal_3d.h:308-322).
struct RadiusCov {
  static constexpr int QPL = 1;
  float r2, kx, ky, kz;
  double a[9];
  uint32_t cnt;
  __device__ __forceinline__ float worst(int) const { return r2; }
  __device__ __forceinline__ void add(bool in, float px, float py, float pz) {
    const float x = in ? __fsub_rn(px, kx) : 0.0f, y = in ? __fsub_rn(py, ky) : 0.0f, z = in ? __fsub_rn(pz, kz) : 0.0f;
    a[0] += double(__fmul_rn(x, x));
    a[1] += double(__fmul_rn(x, y));
    a[2] += double(__fmul_rn(x, z));
    a[3] += double(__fmul_rn(y, y));
    a[4] += double(__fmul_rn(y, z));
    a[5] += double(__fmul_rn(z, z));
    a[6] += double(x);
    a[7] += double(y);
    a[8] += double(z);
    cnt += in ? 1u : 0u;
  }
  __device__ __forceinline__ void leaf(const float* l, uint32_t, const float* qx, const float* qy, const float* qz) {
    const v2f qx2 = {qx[0], qx[0]}, qy2 = {qy[0], qy[0]}, qz2 = {qz[0], qz[0]};
    const v2f* p = reinterpret_cast<const v2f*>(l);
#pragma unroll 1  // (unrolled, the scheduler interleaves the eight pairs' double sums and spills 200 registers)
    for (int j = 0; j < LEAF / 2; ++j) {
      const v2f r = pair_dist(l, j, qx2, qy2, qz2);
      const bool in0 = r.x < r2, in1 = r.y < r2;
      if (__builtin_amdgcn_ballot_w64(in0 || in1) != 0) {  // wave-uniform: most pairs of a leaf are outside every lane's ball
        const v2f px = p[j], py = p[LEAF / 2 + j], pz = p[LEAF + j];
        add(in0, px.x, py.x, pz.x);
        add(in1, px.y, py.y, pz.y);
      }
    }
  }
};

// `q`: the queries in kd order (the index's own points for search surface == input); BYSLOT: the result goes to
// out[q[i].w] (queries of another cloud, Feature::setSearchSurface) instead of out[i].
template <bool BYSLOT>
__global__ __launch_bounds__(BLOCK, 4) void normals_radius_kernel(IndexView ix, const float4* __restrict__ q, uint32_t nq,
                                                                  float r2, float vx, float vy, float vz,
                                                                  float4* __restrict__ nrm,
                                                                  unsigned long long* __restrict__ nan_count) {
  __shared__ WaveLdsBoxT<LEAF_BATCH * LEAF_FLOATS * 4> wl_s[WAVES_PER_BLOCK];
  __shared__ Box topbox_s[TOPCACHE_BOXES];
  load_top_cache(ix, topbox_s);
  const int lane = threadIdx.x & (WAVE - 1);
  const uint32_t ngroups = (nq + WAVE - 1) / WAVE;
  const GroupSchedule sched(ngroups);
  TraverseStats ts;
  GroupFeed feed(sched, ix.sched_ctr);
  uint32_t nans = 0;
  for (uint32_t gl = feed.first(sched); gl != GroupFeed::END; gl = feed.advance()) {
    const uint32_t g = sched.global(gl);
    if (g >= ngroups) break;
    feed.ahead(gl);
    const uint32_t i = g * WAVE + lane;
    float4 p = make_float4(0, 0, 0, 0);
    const bool real = i < nq;
    if (real) p = q[i];
    const bool vv[1] = {real && isfinite(p.x) && isfinite(p.y) && isfinite(p.z)};
    const float qx[1] = {p.x}, qy[1] = {p.y}, qz[1] = {p.z};
    RadiusCov pol;
    pol.r2 = r2;
    pol.kx = p.x;
    pol.ky = p.y;
    pol.kz = p.z;
#pragma unroll
    for (int t = 0; t < 9; ++t) pol.a[t] = 0.0;
    pol.cnt = 0;
    traverse(ix, qx, qy, qz, vv, pol, wl_s[threadIdx.x / WAVE], topbox_s, ts);
    if (!real) continue;
    const float qnan = __builtin_nanf("");
    float4 out = make_float4(qnan, qnan, qnan, qnan);
    if (vv[0] && pol.cnt >= 3) {
      Cov cv;
#pragma unroll
      for (int t = 0; t < 9; ++t) cv.a[t] = float(pol.a[t]);
      float cov[9];
      cv.finish(int(pol.cnt), cov);
      float nx, ny, nz, curv;
      solve_plane(cov, nx, ny, nz, curv);
      flip_to_viewpoint(p.x, p.y, p.z, vx, vy, vz, nx, ny, nz);
      out = make_float4(nx, ny, nz, curv);
    } else {
      ++nans;
    }
    nrm[BYSLOT ? __float_as_uint(p.w) : i] = out;
  }
  const unsigned long long any = __builtin_amdgcn_ballot_w64(nans != 0);
  if (any != 0 && nans != 0) atomicAdd(nan_count, (unsigned long long)nans);
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
void preload_radius_kernels() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(radius_emit_kernel));
}
}  // namespace pclhip

using namespace pclhip;

extern "C" pclhip_status pclhip_radius_search(pclhip_index* ix, const void* queries, size_t stride, uint64_t nq,
                                              double radius, uint32_t max_nn, uint64_t* out_offsets, int32_t* out_idx,
                                              float* out_d2, uint64_t capacity, uint64_t* out_total) {
  if (!ix || !out_offsets || !out_total) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = ix->ctx;
  std::lock_guard<std::recursive_mutex> api_lock(ctx->api_mutex);  // safe under concurrent callers (pclhip.h)
  *out_total = 0;
  PCLHIP_REQUIRE(ctx, stride >= 12 && stride % 4 == 0, "stride must be a multiple of 4 and >= 12 bytes");
  PCLHIP_REQUIRE(ctx, nq < 0x7FFFFFFFull, "too many queries");
  PCLHIP_REQUIRE(ctx, !is_device_pointer(out_offsets), "out_offsets must be host memory");
  out_offsets[0] = 0;
  if (nq == 0) return PCLHIP_OK;
  PCLHIP_REQUIRE(ctx, queries != nullptr, "null buffer");
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  Guard g;
  g.ctx = ctx;
  const void* dq = nullptr;
  void* owned = nullptr;
  pclhip_status st = to_device(ctx, queries, size_t(nq) * stride, &dq, &owned);
  if (st != PCLHIP_OK) return st;
  g.p.push_back(owned);
  float4* qs = nullptr;
  PCLHIP_CHECK_HIP(ctx, g.alloc(&qs, size_t(nq) * sizeof(float4)));
  uint32_t nf = 0;
  float lo[3], hi[3];
  st = spatial_order(ctx, dq, stride, nq, nullptr, 0, qs, uint32_t(nq), &nf, lo, hi, true, nullptr,
                     ix->scaled ? ix->scale : nullptr);
  if (st != PCLHIP_OK) return st;
  const float r2 = float(radius * radius);  // kdtree_flann.hpp:398
  const IndexView v = ix->view();
  const uint32_t n = uint32_t(nq);
  // a batch that is sparse against the index: fewer queries per wavefront (api.hip: sparse_layout); the padding slots count
  // into counts[n] and fill nothing
  const float4* q_run = qs;
  uint32_t n_run = n;
  {
    float4* qe = nullptr;
    st = sparse_layout(ctx, qs, nq, ix->n, &qe, &n_run);
    if (st != PCLHIP_OK) return st;
    if (qe != nullptr) {
      g.p.push_back(qe);
      q_run = qe;
    }
  }
  const uint32_t ngroups = (n_run + WAVE - 1) / WAVE;
  int grid = int((ngroups + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
  const int cap = ctx->num_cus * 4;
  if (grid > cap) grid = cap;
  if (grid < 1) grid = 1;
  uint32_t* counts = nullptr;
  unsigned long long *wide = nullptr, *full_off = nullptr, *out_off = nullptr;
  PCLHIP_CHECK_HIP(ctx, g.alloc(&counts, size_t(n + 1) * 4));
  PCLHIP_CHECK_HIP(ctx, g.alloc(&wide, size_t(n + 1) * 8));
  PCLHIP_CHECK_HIP(ctx, g.alloc(&full_off, size_t(n + 1) * 8));
  PCLHIP_CHECK_HIP(ctx, g.alloc(&out_off, size_t(n + 1) * 8));
  PCLHIP_LAUNCH_FED(ctx, radius_kernel<false>, dim3(grid), dim3(BLOCK), 0, s, v, q_run, n_run, r2, counts,
                     (const unsigned long long*)nullptr, (uint64_t*)nullptr);
  // exclusive scans of the full counts (segment starts) and of the clamped counts (output CSR)
  launch_exclusive_scan_u64(s, counts, n, 0u, wide, full_off);
  launch_exclusive_scan_u64(s, counts, n, max_nn, wide, out_off);
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(out_offsets, out_off, size_t(n + 1) * 8, hipMemcpyDeviceToHost, s));
  unsigned long long full_total = 0;
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(&full_total, full_off + n, 8, hipMemcpyDeviceToHost, s));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
  const uint64_t total = out_offsets[n];
  *out_total = total;
  if (total == 0) return PCLHIP_OK;
  if (!out_idx || !out_d2 || capacity < total) {
    set_error(ctx, "radius search: output capacity too small (out_total holds the required size)");
    return PCLHIP_ERR_OVERFLOW;
  }
  uint64_t* k0 = nullptr;
  PCLHIP_CHECK_HIP(ctx, g.alloc(&k0, size_t(full_total) * 8));
  PCLHIP_LAUNCH_FED(ctx, radius_kernel<true>, dim3(grid), dim3(BLOCK), 0, s, v, q_run, n_run, r2, counts, full_off, k0);
  uint32_t* long_list = nullptr;
  PCLHIP_CHECK_HIP(ctx, g.alloc(&long_list, size_t(n + 1) * 4));  // [n]: the counter
  launch_segmented_sort_u64(s, ctx->num_cus, k0, full_off, 0ull, 0u, n, long_list, long_list + n);
  int32_t* d_idx = out_idx;
  float* d_d2 = out_d2;
  const bool idx_dev = is_device_pointer(out_idx), d2_dev = is_device_pointer(out_d2);
  if (!idx_dev) PCLHIP_CHECK_HIP(ctx, g.alloc(&d_idx, size_t(total) * 4));
  if (!d2_dev) PCLHIP_CHECK_HIP(ctx, g.alloc(&d_d2, size_t(total) * 4));
  hipLaunchKernelGGL(radius_emit_kernel, dim3(n), dim3(64), 0, s, k0, full_off, out_off, n, d_idx, d_d2);
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  if (!idx_dev) PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(out_idx, d_idx, size_t(total) * 4, hipMemcpyDeviceToHost, s));
  if (!d2_dev) PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(out_d2, d_d2, size_t(total) * 4, hipMemcpyDeviceToHost, s));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
  return PCLHIP_OK;
}

// NormalEstimation::setRadiusSearch path: ONE traversal per query group, the covariance accumulated inside it.
static pclhip_status normals_radius_impl(pclhip_index* ix, const float4* queries, uint32_t nq, double radius, const float vp[3],
                                         float4* out, uint64_t* nan_count);

pclhip_status pclhip::launch_normals_radius(pclhip_index* ix, double radius, const float vp[3], uint64_t* nan_count) {
  return normals_radius_impl(ix, nullptr, ix->n, radius, vp, nullptr, nan_count);
}

pclhip_status pclhip::launch_normals_radius_at(pclhip_index* ix, const float4* queries_sorted, uint32_t nq, double radius,
                                               const float vp[3], float4* out, uint64_t* nan_count) {
  return normals_radius_impl(ix, queries_sorted, nq, radius, vp, out, nan_count);
}

// queries == nullptr: the index's own points, normals kept in the index (search surface == input); else `nq` queries
// in kd order with w = output slot, normals to out[slot]
static pclhip_status normals_radius_impl(pclhip_index* ix, const float4* queries, uint32_t nq, double radius, const float vp[3],
                                         float4* out, uint64_t* nan_count) {
  pclhip_ctx* ctx = ix->ctx;
  hipStream_t s = ctx->stream;
  Guard g;
  g.ctx = ctx;
  const bool self = queries == nullptr;
  if (self) {
    if (!ix->nrm) PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &ix->nrm, size_t(ix->n_pad > 0 ? ix->n_pad : 1) * sizeof(float4)));
    PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(ix->nrm, 0xFF, size_t(ix->n_pad) * sizeof(float4), s));  // NaN pads
  }
  if (nan_count) *nan_count = 0;
  uint32_t n = nq;
  const float4* q = self ? ix->pts : queries;
  float4* dst = self ? ix->nrm : out;
  if (n == 0) {
    if (self) ix->has_normals = true;
    return PCLHIP_OK;
  }
  // few queries on a large surface: fewer of them per wavefront (api.hip: sparse_layout).  The padding slots write their NaN
  // row to a dump row behind the results and are counted as NaN rows by the kernel: both taken back below.
  uint32_t pads = 0;
  if (!self) {
    float4* qe = nullptr;
    uint32_t n_run = n;
    const pclhip_status sl = sparse_layout(ctx, queries, nq, ix->n, &qe, &n_run);
    if (sl != PCLHIP_OK) return sl;
    if (qe != nullptr) {
      g.p.push_back(qe);
      PCLHIP_CHECK_HIP(ctx, g.alloc(&dst, (size_t(nq) + 1) * sizeof(float4)));
      q = qe;
      pads = n_run - nq;
      n = n_run;
    }
  }
  const float r2 = float(radius * radius);  // kdtree_flann.hpp:398
  const IndexView v = ix->view();
  const uint32_t ngroups = (n + WAVE - 1) / WAVE;
  int grid = int((ngroups + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
  const int cap = ctx->num_cus * 4;
  if (grid > cap) grid = cap;
  if (grid < 1) grid = 1;
  unsigned long long* d_nan = nullptr;
  PCLHIP_CHECK_HIP(ctx, g.alloc(&d_nan, 8));
  PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(d_nan, 0, 8, s));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, s);
  if (self)
    PCLHIP_LAUNCH_FED(ctx, normals_radius_kernel<false>, dim3(grid), dim3(BLOCK), 0, s, v, q, n, r2, vp[0], vp[1], vp[2], dst,
                       d_nan);
  else
    PCLHIP_LAUNCH_FED(ctx, normals_radius_kernel<true>, dim3(grid), dim3(BLOCK), 0, s, v, q, n, r2, vp[0], vp[1], vp[2], dst,
                       d_nan);
  (void)hipEventRecord(e1, s);
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  unsigned long long h = 0;
  if (dst != out && !self) PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, dst, size_t(nq) * sizeof(float4), hipMemcpyDeviceToDevice, s));
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(&h, d_nan, 8, hipMemcpyDeviceToHost, s));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
  float ms = 0;
  if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) ix->last_kernel_ms = ms;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (nan_count) *nan_count = h - pads;
  if (self) ix->has_normals = true;
  return PCLHIP_OK;
}
