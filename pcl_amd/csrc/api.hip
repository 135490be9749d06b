// api.hip -- the C ABI (include/pclhip.h) over the HIP kernels, plus the host ICP loop.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pclhip_internal.hpp"
#include "device_scan.hpp"

using namespace pclhip;

namespace {
thread_local std::string g_last_error;
}

namespace pclhip {

void set_error(pclhip_ctx* ctx, const std::string& msg) {
  g_last_error = msg;
  if (ctx) ctx->last_error = msg;
}

bool is_device_pointer(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // unregistered host memory: clear the sticky error
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// Bytes from the first to the last one read of n records `stride` apart of which `elem` bytes each are read.  Callers hand
// over pointers INTO their records (the normals of pcl::PointNormal: record + 16); copying n * stride from there would read
// past the end of their array (found by the sanitizer run of the C++ binding on the emulation, tests/test_wavesim.py).
static inline size_t strided_span(uint64_t n, size_t stride, size_t elem) {
  return n == 0 ? 0 : size_t(n - 1) * stride + elem;
}

pclhip_status to_device(pclhip_ctx* ctx, const void* p, size_t bytes, const void** dev, void** owned) {
  *owned = nullptr;
  *dev = nullptr;
  if (bytes == 0 || p == nullptr) return PCLHIP_OK;
  if (is_device_pointer(p)) {
    *dev = p;
    return PCLHIP_OK;
  }
  void* d = nullptr;
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &d, bytes));   // every caller releases `owned` with dev_free
  hipError_t e = hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e != hipSuccess) {
    dev_free(ctx, d);
    PCLHIP_CHECK_HIP(ctx, e);
  }
  *owned = d;
  *dev = d;
  return PCLHIP_OK;
}

pclhip_status ensure_scratch(pclhip_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->scratch_bytes) return PCLHIP_OK;
  if (ctx->scratch) {  // stream-ordered like every other block of the context: no synchronisation needed
    dev_free(ctx, ctx->scratch);
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
  }
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &ctx->scratch, bytes));
  ctx->scratch_bytes = bytes;
  return PCLHIP_OK;
}

// ---- context-cached device memory -----------------------------------------------------------------
namespace {
constexpr size_t ARENA_ALIGN = 256;
// first fit in the arena's free ranges (the caller holds cache_mutex); nullptr when nothing fits
void* arena_take(pclhip_ctx* ctx, size_t bytes) {
  for (auto it = ctx->arena_free.begin(); it != ctx->arena_free.end(); ++it) {
    if (it->second < bytes) continue;
    const size_t off = it->first, rest = it->second - bytes;
    ctx->arena_free.erase(it);
    if (rest > 0) ctx->arena_free.emplace(off + bytes, rest);
    return ctx->arena + off;
  }
  return nullptr;
}
void arena_give(pclhip_ctx* ctx, void* p, size_t bytes) {
  size_t off = size_t(static_cast<char*>(p) - ctx->arena);
  auto next = ctx->arena_free.lower_bound(off);
  if (next != ctx->arena_free.begin()) {  // merge with the range before
    auto prev = std::prev(next);
    if (prev->first + prev->second == off) {
      off = prev->first;
      bytes += prev->second;
      ctx->arena_free.erase(prev);
    }
  }
  if (next != ctx->arena_free.end() && off + bytes == next->first) {  // ... and the one after
    bytes += next->second;
    ctx->arena_free.erase(next);
  }
  ctx->arena_free.emplace(off, bytes);
}
bool in_arena(const pclhip_ctx* ctx, const void* p) {
  return ctx->arena != nullptr && static_cast<const char*>(p) >= ctx->arena &&
         static_cast<const char*>(p) < ctx->arena + ctx->arena_bytes;
}
}  // namespace

pclhip_status reserve_arena(pclhip_ctx* ctx, size_t bytes) {
  std::lock_guard<std::mutex> lock(ctx->cache_mutex);
  if (ctx->arena != nullptr || bytes == 0) return PCLHIP_OK;
  bytes = (bytes + ARENA_ALIGN - 1) & ~(ARENA_ALIGN - 1);
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return PCLHIP_ERR_HIP;  // not fatal: allocations go through hipMalloc + the block cache
  }
  ctx->arena = static_cast<char*>(p);
  ctx->arena_bytes = bytes;
  ctx->arena_free.clear();
  ctx->arena_free.emplace(size_t(0), bytes);
  return PCLHIP_OK;
}

// Device bytes a registration pipeline holds per point of its largest cloud: the index (points, SoA copy, normals,
// boxes, discs, rank: ~60), the build's and the source ordering's scratch (~100), the source arrays of a registration
// (copies, matches, distances: ~50) and a VoxelGrid pass (~60) -- 288 with slack.  Option "arena_mb" overrides (0: none).
void dev_reserve_for_points(pclhip_ctx* ctx, uint64_t points) {
  if (ctx == nullptr || points < 1000000ull) return;
  {
    std::lock_guard<std::mutex> lock(ctx->cache_mutex);  // (two threads of one context may bring their first clouds at once)
    if (ctx->arena_tried || ctx->arena != nullptr) return;
    ctx->arena_tried = true;
  }
  size_t want = size_t(points) * 288;
  if (ctx->opt_arena_mb >= 0) want = size_t(ctx->opt_arena_mb) << 20;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && want > free_b / 2) want = free_b / 2;
  if (want > 0) (void)reserve_arena(ctx, want);
}

hipError_t pinned_malloc(pclhip_ctx* ctx, void** p, size_t bytes) {
  {
    std::lock_guard<std::mutex> lock(ctx->cache_mutex);
    for (size_t i = 0; i < ctx->pinned_cache.size(); ++i)
      if (ctx->pinned_cache[i].second == bytes) {
        *p = ctx->pinned_cache[i].first;
        ctx->pinned_cache.erase(ctx->pinned_cache.begin() + long(i));
        return hipSuccess;
      }
  }
  return hipHostMalloc(p, bytes, hipHostMallocDefault);
}

void pinned_free(pclhip_ctx* ctx, void* p, size_t bytes) {
  if (!p) return;
  std::lock_guard<std::mutex> lock(ctx->cache_mutex);
  if (ctx->pinned_cache.size() < 64 && bytes <= (size_t(1) << 20)) {
    ctx->pinned_cache.emplace_back(p, bytes);
    return;
  }
  (void)hipHostFree(p);
}

hipError_t dev_malloc(pclhip_ctx* ctx, void** p, size_t bytes) {
  *p = nullptr;
  if (bytes == 0) bytes = 16;
  if (ctx == nullptr) return hipMalloc(p, bytes);
  {
    std::lock_guard<std::mutex> lock(ctx->cache_mutex);
    if (ctx->arena != nullptr) {
      const size_t rounded = (bytes + ARENA_ALIGN - 1) & ~(ARENA_ALIGN - 1);
      if (void* a = arena_take(ctx, rounded)) {
        *p = a;
        ctx->live[a] = rounded;
        return hipSuccess;
      }
    }
    // best fit among the free blocks, but never one more than twice the size asked for
    size_t best = ctx->cache.size();
    for (size_t i = 0; i < ctx->cache.size(); ++i) {
      const size_t b = ctx->cache[i].second;
      if (b >= bytes && b <= 2 * bytes + (size_t(1) << 20) && (best == ctx->cache.size() || b < ctx->cache[best].second)) best = i;
    }
    if (best != ctx->cache.size()) {
      *p = ctx->cache[best].first;
      ctx->live[*p] = ctx->cache[best].second;
      ctx->cached_bytes -= ctx->cache[best].second;
      ctx->cache.erase(ctx->cache.begin() + long(best));
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {  // out of memory: give the cached blocks back and try once more
    (void)hipGetLastError();
    dev_cache_release(ctx);
    e = hipMalloc(p, bytes);
  }
  if (e != hipSuccess) {  // still: an arena nobody holds a block of goes back to the device too (another context or
    (void)hipGetLastError();  // process on this GPU, or one allocation larger than what the arena left free)
    char* idle = nullptr;
    {
      std::lock_guard<std::mutex> lock(ctx->cache_mutex);
      if (ctx->arena != nullptr && ctx->arena_free.size() == 1 && ctx->arena_free.begin()->first == 0 &&
          ctx->arena_free.begin()->second == ctx->arena_bytes) {
        idle = ctx->arena;
        ctx->arena = nullptr;
        ctx->arena_bytes = 0;
        ctx->arena_free.clear();
      }
    }
    if (idle != nullptr) {
      (void)hipStreamSynchronize(ctx->stream);  // blocks handed back to the arena may still be read by queued work
      (void)hipFree(idle);
      e = hipMalloc(p, bytes);
    }
  }
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lock(ctx->cache_mutex);
    ctx->live[*p] = bytes;
  }
  return e;
}

void dev_free(pclhip_ctx* ctx, void* p) {
  if (p == nullptr) return;
  if (ctx != nullptr) {
    std::lock_guard<std::mutex> lock(ctx->cache_mutex);
    const auto it = ctx->live.find(p);
    if (it != ctx->live.end()) {
      const size_t bytes = it->second;
      ctx->live.erase(it);
      if (in_arena(ctx, p)) {  // all work of the context is ordered on its stream: the range can be handed out again
        arena_give(ctx, p, bytes);
        return;
      }
      if (ctx->cached_bytes + bytes <= ctx->cache_limit) {
        ctx->cache.emplace_back(p, bytes);
        ctx->cached_bytes += bytes;
        return;
      }
    } else {
      if (in_arena(ctx, p)) return;  // released twice
      for (const auto& b : ctx->cache)
        if (b.first == p) return;  // released twice: it already sits in the cache
    }
  }
  (void)hipFree(p);  // not from the cache (or the cache is full): the plain, synchronising free
}

void dev_cache_release(pclhip_ctx* ctx) {
  std::vector<std::pair<void*, size_t>> blocks;
  {
    std::lock_guard<std::mutex> lock(ctx->cache_mutex);
    blocks.swap(ctx->cache);
    ctx->cached_bytes = 0;
  }
  if (!blocks.empty()) (void)hipStreamSynchronize(ctx->stream);
  for (auto& b : blocks) (void)hipFree(b.first);
}

namespace {

struct DeviceGuard {  // releases staged copies and temporaries on scope exit (back to the context's cache)
  pclhip_ctx* ctx;
  std::vector<void*> ptrs;
  explicit DeviceGuard(pclhip_ctx* c) : ctx(c) {}
  ~DeviceGuard() {
    for (void* p : ptrs)
      if (p) dev_free(ctx, p);
  }
  void add(void* p) { ptrs.push_back(p); }
};

// write `bytes_per_rec` bytes per record from a dense device array to a strided user buffer
pclhip_status copy_out_strided(pclhip_ctx* ctx, void* user, size_t stride, const void* dev_dense, size_t rec_bytes,
                               uint64_t n) {
  if (n == 0) return PCLHIP_OK;
  const hipMemcpyKind kind = is_device_pointer(user) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  PCLHIP_CHECK_HIP(ctx, hipMemcpy2DAsync(user, stride, dev_dense, rec_bytes, rec_bytes, n, kind, ctx->stream));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PCLHIP_OK;
}

__global__ void scatter_normals_kernel(const float4* __restrict__ nrm_sorted, const uint32_t* __restrict__ rank,
                                       uint64_t n_orig, float4* __restrict__ out) {
  const uint64_t i = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
  if (i >= n_orig) return;
  const uint32_t r = rank[i];
  const float qn = __builtin_nanf("");
  out[i] = (r == NO_INDEX) ? make_float4(qn, qn, qn, qn) : nrm_sorted[r];
}

// whole output records (pcl::Normal and the like: the reference value-initialises them before it writes the normal):
// zeros, the normal at normal_off, the curvature at curvature_off; NaN for dropped points
__global__ void normal_records_kernel(const float4* __restrict__ nrm_sorted, const uint32_t* __restrict__ rank, uint64_t n_orig,
                                      char* __restrict__ out, uint32_t record_bytes, uint32_t normal_off,
                                      uint32_t curvature_off) {
  const uint64_t i = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
  if (i >= n_orig) return;
  const uint32_t r = rank[i];
  const float qn = __builtin_nanf("");
  const float4 v = (r == NO_INDEX) ? make_float4(qn, qn, qn, qn) : nrm_sorted[r];
  float* rec = reinterpret_cast<float*>(out + i * record_bytes);
  for (uint32_t w = 0; w < record_bytes / 4; ++w) rec[w] = 0.0f;
  float* n = reinterpret_cast<float*>(out + i * record_bytes + normal_off);
  n[0] = v.x; n[1] = v.y; n[2] = v.z;
  *reinterpret_cast<float*>(out + i * record_bytes + curvature_off) = v.w;
}

__global__ void gather_normals_kernel(const void* normals, size_t stride, const float4* __restrict__ pts_sorted,
                                      uint32_t n, uint32_t n_pad, float4* __restrict__ nrm_sorted) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_pad) return;
  const float qn = __builtin_nanf("");
  if (j < n) {
    const uint32_t rec = __float_as_uint(pts_sorted[j].w);
    const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(normals) + size_t(rec) * stride);
    nrm_sorted[j] = make_float4(p[0], p[1], p[2], 0.0f);
  } else {
    nrm_sorted[j] = make_float4(qn, qn, qn, qn);
  }
}

struct Mat44 {
  float m[16];
};

// *bad = 1 if any sel[i] is outside [0, n)
__global__ void check_indices_kernel(const int32_t* __restrict__ sel, uint64_t m, uint64_t n, int* __restrict__ bad) {
  const uint64_t i = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
  if (i < m) {
    const int32_t v = sel[i];
    if (v < 0 || uint64_t(v) >= n) *bad = 1;
  }
}

__global__ void transform_cloud_kernel(Mat44 T, int order, const void* in, void* out, size_t stride, uint64_t n,
                                       size_t nrm_off) {
  const uint64_t i = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(in) + i * stride);
  float* o = reinterpret_cast<float*>(reinterpret_cast<char*>(out) + i * stride);
  const float x = p[0], y = p[1], z = p[2];
  if (!(isfinite(x) && isfinite(y) && isfinite(z))) return;
  float r[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float* row = T.m + 4 * k;
    if (order == 0)
      r[k] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(row[0], x), __fmul_rn(row[1], y)), __fmul_rn(row[2], z)),
                       __fmul_rn(row[3], 1.0f));
    else
      r[k] = __fadd_rn(__fmul_rn(row[0], x), __fadd_rn(__fmul_rn(row[1], y), __fadd_rn(__fmul_rn(row[2], z), row[3])));
  }
  o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
  if (nrm_off) {
    const float* np = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + nrm_off);
    float* no = reinterpret_cast<float*>(reinterpret_cast<char*>(o) + nrm_off);
    const float a = np[0], b = np[1], c = np[2];
    if (order == 0 && !(isfinite(a) && isfinite(b) && isfinite(c))) return;
    float s[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float* row = T.m + 4 * k;
      if (order == 0)
        s[k] = __fadd_rn(__fadd_rn(__fmul_rn(row[0], a), __fmul_rn(row[1], b)), __fmul_rn(row[2], c));
      else
        s[k] = __fadd_rn(__fmul_rn(row[0], a), __fadd_rn(__fmul_rn(row[1], b), __fmul_rn(row[2], c)));
    }
    no[0] = s[0]; no[1] = s[1]; no[2] = s[2];
  }
}

// dense per-original-source arrays of the last iteration's matches
__global__ void scatter_matches_kernel(const float4* __restrict__ cur, const uint32_t* __restrict__ match,
                                       const float* __restrict__ d2, const uint8_t* __restrict__ keep, uint32_t n,
                                       int32_t* __restrict__ out_m, float* __restrict__ out_d) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t oq = __float_as_uint(cur[i].w);
  const uint32_t m = match[i];
  const bool kept = m != NO_INDEX && (keep == nullptr || keep[i]);
  out_m[oq] = kept ? int32_t(m) : -1;
  out_d[oq] = d2[i];
}

// pcl::Correspondence records (index_query, index_match, distance: 12 bytes) of the kept pairs, ascending by query
__global__ void match_flag_kernel(const int32_t* __restrict__ m, uint64_t no, uint32_t* __restrict__ flag) {
  const uint64_t i = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
  if (i < no) flag[i] = m[i] >= 0 ? 1u : 0u;
}
struct CorrRecord {
  int32_t q, m;
  float d;
};
__global__ void match_emit_kernel(const int32_t* __restrict__ m, const float* __restrict__ d, const uint32_t* __restrict__ pos,
                                  uint64_t no, CorrRecord* __restrict__ out) {
  const uint64_t i = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
  if (i < no && m[i] >= 0) out[pos[i]] = CorrRecord{int32_t(i), m[i], d[i]};
}

float float_at_most(double v) {  // largest float <= v  (v >= 0)
  if (!(v < double(FLT_MAX))) return FLT_MAX;
  float f = float(v);
  if (double(f) > v) f = std::nextafterf(f, 0.0f);
  return f;
}

}  // namespace
}  // namespace pclhip

namespace pclhip {
// The runtime loads a translation unit's code object at the first use of one of its kernels: that is a context's
// business, not the first timed launch's (the driver's fresh box showed the first index build at 12x the steady one).
void preload_code_objects(pclhip_ctx* ctx) {
  preload_search_kernels(ctx);
  preload_index_build_kernels();
  preload_voxelgrid_kernels();
  preload_rejector_kernels();
  preload_radius_kernels();
  preload_lane_kernels();
  (void)hipGetLastError();
}
}  // namespace pclhip

pclhip_status pclhip::sharded_filters_ok(pclhip_icp* icp) {
  pclhip_ctx* ctx = icp->ctx;
  if (icp_is_sharded(icp)) {
    // Per-pair filters (Distance) commute with the sharding; MedianDistance / Trimmed take ONE cloud-global order
    // statistic, which the ranks find together (the histograms of the selection are all-reduced, rejectors.hip).  The
    // reciprocal test needs the whole moved source on the rank: fine when the TARGET is sharded (a region is set: the
    // source is replicated; the served-group lists stand aside), not when the source is cut into slabs.  OneToOne
    // resolves conflicts between pairs that different ranks serve (a target point in two halos): the per-target minimum
    // keys are reduced over the ranks (ncclMin on 64-bit keys) -- with the native communicator and a sharded TARGET, where
    // every rank holds the whole source and query indices are global; with source slabs the indices are rank-local.
    for (const pclhip_rejector& r : icp->rejectors) {
      if (r.kind == PCLHIP_REJ_ONE_TO_ONE && (icp->comm == nullptr || icp->region.on == 0)) {
        set_error(ctx, "the OneToOne rejector under multi-GPU iterations needs the native communicator (pclhip_icp_set_comm) "
                       "and a sharded target (pclhip_icp_set_region): with source slabs or an all-reduce hook its conflicts "
                       "cannot be resolved");
        return PCLHIP_ERR_STATE;
      }
    }
    if (icp->reciprocal && icp->region.on == 0) {
      set_error(ctx, "reciprocal correspondences need the whole source on every rank: shard the target (pclhip_icp_set_region), "
                     "not the source");
      return PCLHIP_ERR_STATE;
    }
  }
  return PCLHIP_OK;
}

extern "C" {

// The CPU emulation of the test tier (tests/wavesim) says what it is -- pcl_amd/_lib.py refuses it unless asked --: its runtime
// DEFINES this function, the product does not (a weak reference, null here): a link-time hook, no conditional compilation.
__attribute__((weak)) const char* pclhip_emulation_banner(int verify_bounds);
const char* pclhip_version(void) {
  // (search.hip says whether it was compiled with -DPCLHIP_VERIFY_BOUNDS: scripts/build_variant.sh rebuilds that unit only)
  const bool vb = pclhip::search_built_with_verify_bounds();
  if (&pclhip_emulation_banner != nullptr) return pclhip_emulation_banner(vb ? 1 : 0);
  return vb ? "pclhip 0.1 (gfx950, verify-bounds build: test infrastructure)" : "pclhip 0.1 (gfx950)";
}

const char* pclhip_last_error(const pclhip_ctx* ctx) { return ctx ? ctx->last_error.c_str() : g_last_error.c_str(); }

static pclhip_status ctx_create_impl(int device, void* stream, bool adopt, pclhip_ctx** out);
static pclhip_status transform_cloud_impl(pclhip_ctx* ctx, const float* T, int order, const void* in, const void* resident,
                                          void* out, size_t stride, uint64_t n, size_t normals_offset_bytes);

pclhip_status pclhip_ctx_create(int device, void* stream, pclhip_ctx** out) {
  return ctx_create_impl(device, stream, stream != nullptr, out);
}

pclhip_status pclhip_ctx_create_on_stream(int device, void* stream, pclhip_ctx** out) {
  return ctx_create_impl(device, stream, true, out);
}

void* pclhip_ctx_stream(const pclhip_ctx* ctx) { return ctx ? static_cast<void*>(ctx->stream) : nullptr; }

static pclhip_status ctx_create_impl(int device, void* stream, bool adopt, pclhip_ctx** out) {
  if (!out) return PCLHIP_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    set_error(nullptr, "no HIP device visible");
    return PCLHIP_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= count) {
    set_error(nullptr, "device ordinal out of range");
    return PCLHIP_ERR_INVALID;
  }
  pclhip_ctx* ctx = new pclhip_ctx();
  ctx->device = device;
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(device));
  hipDeviceProp_t prop;
  PCLHIP_CHECK_HIP(ctx, hipGetDeviceProperties(&prop, device));
  ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (adopt) {
    ctx->stream = static_cast<hipStream_t>(stream);  // may be the null (legacy default) stream
  } else {
    PCLHIP_CHECK_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
  }
  {  // never hold back more than a quarter of the device's memory
    const size_t quarter = size_t(prop.totalGlobalMem) / 4;
    if (ctx->cache_limit > quarter) ctx->cache_limit = quarter;
  }
  PCLHIP_CHECK_HIP(ctx, hipMalloc(&ctx->sched_ctr, pclhip::SCHED_CTR_BYTES));
  PCLHIP_CHECK_HIP(ctx, hipMemset(ctx->sched_ctr, 0, pclhip::SCHED_CTR_BYTES));
  preload_code_objects(ctx);
  *out = ctx;
  return PCLHIP_OK;
}


pclhip_status pclhip_ctx_reserve(pclhip_ctx* ctx, uint64_t bytes) {
  if (!ctx) return PCLHIP_ERR_INVALID;
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  ctx->arena_tried = true;
  const pclhip_status st = pclhip::reserve_arena(ctx, size_t(bytes));
  if (st != PCLHIP_OK) pclhip::set_error(ctx, "could not reserve the arena (allocations fall back to hipMalloc)");
  return st;
}

pclhip_status pclhip_ctx_set_option(pclhip_ctx* ctx, const char* name, double value) {
  if (!ctx || !name) return PCLHIP_ERR_INVALID;
  if (!(value >= 0.0) || value > 1e9) {
    pclhip::set_error(ctx, "option values are non-negative numbers");
    return PCLHIP_ERR_INVALID;
  }
  if (!std::strcmp(name, "served_groups")) {
    ctx->opt_served_groups = value != 0.0 ? 1 : 0;
  } else if (!std::strcmp(name, "icp_lookahead")) {
    ctx->opt_lookahead = value > 62.0 ? 62 : int(value);  // the step ring holds 64 records (icp_loop.hip)
  } else if (!std::strcmp(name, "cache_mb")) {
    size_t limit = size_t(value) << 20;
    size_t free_b = 0, total_b = 0;
    if (hipSetDevice(ctx->device) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess && limit > total_b / 4)
      limit = total_b / 4;  // never hold back more than a quarter of the device's memory (as at creation)
    std::lock_guard<std::mutex> lock(ctx->cache_mutex);
    ctx->cache_limit = limit;
  } else if (!std::strcmp(name, "arena_mb")) {
    if (ctx->arena_tried) {
      pclhip::set_error(ctx, "arena_mb: the context's arena exists already (set it before the first large cloud)");
      return PCLHIP_ERR_STATE;
    }
    ctx->opt_arena_mb = (long long)value;
  } else if (!std::strcmp(name, "lane_search")) {
    ctx->opt_lane_search = value != 0.0 ? 1 : 0;
  } else if (!std::strcmp(name, "cell_start")) {
    ctx->opt_cell_start = value != 0.0 ? 1 : 0;
  } else if (!std::strcmp(name, "reseed")) {
    ctx->opt_reseed = value != 0.0 ? 1 : 0;
  } else if (!std::strcmp(name, "standoff_thickness")) {
    ctx->opt_standoff_thickness = float(value);
  } else if (!std::strcmp(name, "standoff_max_mb")) {
    ctx->opt_standoff_max_mb = value < 0.0 ? 0 : int(value);
  } else if (!std::strcmp(name, "lane_max_up")) {
    ctx->opt_lane_max_up = value > 15.0 ? 15 : int(value);
  } else if (!std::strcmp(name, "lane_far")) {
    ctx->opt_lane_far = float(value);
  } else {
    pclhip::set_error(ctx, "unknown option (served_groups, icp_lookahead, cache_mb, arena_mb, cell_start, reseed, lane_search, lane_max_up, lane_far)");
    return PCLHIP_ERR_INVALID;
  }
  return PCLHIP_OK;
}

void pclhip_ctx_destroy(pclhip_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->scratch) dev_free(ctx, ctx->scratch);
  if (ctx->stats) (void)hipFree(ctx->stats);
  if (ctx->sched_ctr) (void)hipFree(ctx->sched_ctr);
  if (ctx->staging) (void)hipFree(ctx->staging);
  dev_cache_release(ctx);
  for (auto& b : ctx->pinned_cache) (void)hipHostFree(b.first);
  ctx->pinned_cache.clear();
  if (ctx->arena) (void)hipFree(ctx->arena);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

pclhip_status pclhip_ctx_stats(pclhip_ctx* ctx, int enable, uint64_t* out) {
  if (!ctx) return PCLHIP_ERR_INVALID;
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (out) {
    if (ctx->stats)
      PCLHIP_CHECK_HIP(ctx, hipMemcpy(out, ctx->stats, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    else
      std::memset(out, 0, 8 * sizeof(uint64_t));
  }
  if (enable) {
    if (!ctx->stats) PCLHIP_CHECK_HIP(ctx, hipMalloc(&ctx->stats, 8 * sizeof(uint64_t)));
    PCLHIP_CHECK_HIP(ctx, hipMemset(ctx->stats, 0, 8 * sizeof(uint64_t)));
  } else if (ctx->stats) {
    (void)hipFree(ctx->stats);
    ctx->stats = nullptr;
  }
  return PCLHIP_OK;
}

pclhip_status pclhip_ctx_synchronize(pclhip_ctx* ctx) {
  if (!ctx) return PCLHIP_ERR_INVALID;
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PCLHIP_OK;
}

// ------------------------------------------------------------------------------------------------
pclhip_status pclhip_index_build(pclhip_ctx* ctx, const void* points, size_t stride, uint64_t n,
                                 const int32_t* indices, uint64_t n_indices, pclhip_index** out) {
  return pclhip_index_build_scaled(ctx, points, stride, n, indices, n_indices, nullptr, out);
}

pclhip_status pclhip_index_build_scaled(pclhip_ctx* ctx, const void* points, size_t stride, uint64_t n,
                                        const int32_t* indices, uint64_t n_indices, const float* scale,
                                        pclhip_index** out) {
  return pclhip_index_build_ex(ctx, points, stride, n, indices, n_indices, scale, 0, out);
}

pclhip_status pclhip_index_build_ex(pclhip_ctx* ctx, const void* points, size_t stride, uint64_t n,
                                    const int32_t* indices, uint64_t n_indices, const float* scale,
                                    size_t normals_offset_bytes, pclhip_index** out) {
  if (!ctx || !out) return PCLHIP_ERR_INVALID;
  PCLHIP_REQUIRE(ctx, normals_offset_bytes == 0 || (normals_offset_bytes % 4 == 0 && normals_offset_bytes + 16 <= stride),
                 "the normals (nx, ny, nz, curvature: 16 bytes) must lie inside the record");
  if (scale) PCLHIP_REQUIRE(ctx, std::isfinite(scale[0]) && std::isfinite(scale[1]) && std::isfinite(scale[2]), "non-finite rescale value");
  *out = nullptr;
  PCLHIP_REQUIRE(ctx, stride >= 12 && stride % 4 == 0, "stride must be a multiple of 4 and >= 12 bytes");
  PCLHIP_REQUIRE(ctx, n < 0x7FFFFFFFull, "cloud too large for int32 indices");
  PCLHIP_REQUIRE(ctx, n == 0 || points != nullptr, "null point buffer");
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  dev_reserve_for_points(ctx, n);  // the context's arena, sized by its first large cloud
  DeviceGuard guard(ctx);
  const void* dpts = nullptr;
  void* owned = nullptr;
  pclhip_status st = to_device(ctx, points, size_t(n) * stride, &dpts, &owned);
  if (st != PCLHIP_OK) return st;
  guard.add(owned);
  const void* dsel = nullptr;
  if (indices) {
    st = to_device(ctx, indices, size_t(n_indices) * sizeof(int32_t), &dsel, &owned);
    if (st != PCLHIP_OK) return st;
    guard.add(owned);
  }
  const uint64_t m = indices ? n_indices : n;
  if (indices && n_indices > 0) {  // untrusted: an index outside the cloud would be a wild device read
    PCLHIP_REQUIRE(ctx, n_indices < 0x7FFFFFFFull, "too many indices");
    int* bad = nullptr;
    PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &bad, sizeof(int)));
    guard.add(bad);
    PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(check_indices_kernel, dim3(unsigned((n_indices + 255) / 256)), dim3(256), 0, ctx->stream,
                       static_cast<const int32_t*>(dsel), n_indices, n, bad);
    int hbad = 0;
    PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PCLHIP_REQUIRE(ctx, hbad == 0, "indices must lie in [0, n)");
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  PCLHIP_CHECK_HIP(ctx, hipEventCreate(&e0));
  if (hipEventCreate(&e1) != hipSuccess) {
    (void)hipEventDestroy(e0);
    set_error(ctx, "hipEventCreate failed");
    return PCLHIP_ERR_HIP;
  }
  pclhip_index* ix = new pclhip_index();
  ix->ctx = ctx;
  ix->n_orig = n;
  if (scale) {
    ix->scaled = true;
    for (int d = 0; d < 3; ++d) ix->scale[d] = scale[d];
  }
  (void)hipEventRecord(e0, ctx->stream);
  const uint32_t cap = uint32_t(((m + LEAF - 1) / LEAF) * LEAF) + LEAF;
  auto fail = [&](pclhip_status s) {
    pclhip_index_destroy(ix);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return s;
  };
  if (dev_malloc(ctx, &ix->pts, size_t(cap) * sizeof(float4)) != hipSuccess ||
      dev_malloc(ctx, &ix->rank, size_t(n > 0 ? n : 1) * sizeof(uint32_t)) != hipSuccess) {
    set_error(ctx, "hipMalloc failed for the index");
    return fail(PCLHIP_ERR_HIP);
  }
  uint32_t nf = 0;
  st = spatial_order(ctx, dpts, stride, n, static_cast<const int32_t*>(dsel), n_indices, ix->pts, cap, &nf, ix->bbox_lo,
                    ix->bbox_hi, false, ix->rank, scale);
  if (st != PCLHIP_OK) return fail(st);
  ix->n = nf;
  ix->n_pad = ((nf + LEAF - 1) / LEAF) * LEAF;
  if (ix->n_pad == 0) ix->n_pad = LEAF;
  st = build_boxes(ix);
  if (st != PCLHIP_OK) return fail(st);
  (void)hipEventRecord(e1, ctx->stream);
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
    set_error(ctx, "index build failed on the device");
    return fail(PCLHIP_ERR_HIP);
  }
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ix->build_ms = ms;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (normals_offset_bytes != 0 && n > 0) {
    // the records are on the device already (staged for the build, or the caller's device buffer): the normals come
    // from the same copy -- a host cloud of PointNormal records is uploaded once, not twice
    st = pclhip_index_set_normals(ix, static_cast<const char*>(dpts) + normals_offset_bytes, stride);
    if (st != PCLHIP_OK) {
      pclhip_index_destroy(ix);
      return st;
    }
  }
  *out = ix;
  return PCLHIP_OK;
}

void pclhip_index_destroy(pclhip_index* ix) {
  if (!ix) return;
  if (ix->ctx) {
    (void)hipSetDevice(ix->ctx->device);
    (void)hipStreamSynchronize(ix->ctx->stream);
  }
  if (ix->pts && !ix->pts_borrowed) (void)dev_free(ix->ctx, ix->pts);
  if (ix->soa) (void)dev_free(ix->ctx, ix->soa);
  if (ix->nrm) (void)dev_free(ix->ctx, ix->nrm);
  if (ix->disc) (void)dev_free(ix->ctx, ix->disc);
  if (ix->rank) (void)dev_free(ix->ctx, ix->rank);
  if (ix->lv_dev) (void)dev_free(ix->ctx, ix->lv_dev);
  if (ix->topcache) (void)dev_free(ix->ctx, ix->topcache);
  if (ix->qbox) ix->box[1] = nullptr;  // the leaf boxes are the front of qbox
  if (ix->qbox) (void)dev_free(ix->ctx, ix->qbox);
  if (ix->qcell) (void)dev_free(ix->ctx, ix->qcell);
  for (int l = 0; l < MAX_LEVELS; ++l)
    if (ix->box[l]) (void)dev_free(ix->ctx, ix->box[l]);
  delete ix;
}

uint64_t pclhip_index_size(const pclhip_index* ix) { return ix ? ix->n : 0; }
double pclhip_index_build_ms(const pclhip_index* ix) { return ix ? ix->build_ms : 0.0; }

namespace {
__global__ void index_order_kernel(const float4* __restrict__ pts, uint32_t n, int32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = int32_t(__float_as_uint(pts[i].w));
}
}  // namespace

pclhip_status pclhip_index_order(pclhip_index* ix, int32_t* out) {
  if (!ix) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = ix->ctx;
  if (ix->n == 0) return PCLHIP_OK;
  PCLHIP_REQUIRE(ctx, out != nullptr, "null buffer");
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  DeviceGuard guard(ctx);
  int32_t* d = out;
  const bool dev = is_device_pointer(out);
  if (!dev) {
    PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &d, size_t(ix->n) * sizeof(int32_t)));
    guard.add(d);
  }
  hipLaunchKernelGGL(index_order_kernel, dim3((ix->n + 255) / 256), dim3(256), 0, ctx->stream, ix->pts, ix->n, d);
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  if (!dev) PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, d, size_t(ix->n) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PCLHIP_OK;
}

pclhip_status pclhip_index_cells(pclhip_index* ix, int level, float* boxes, float* cells, uint64_t capacity, uint64_t* count,
                                 int* top_level) {
  if (!ix) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = ix->ctx;
  if (ix->qcell == nullptr) {
    set_error(ctx, "this index carries no per-lane search structure");
    return PCLHIP_ERR_STATE;
  }
  PCLHIP_REQUIRE(ctx, level >= 0 && level <= ix->qtop, "level out of range");
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t nleaf = ix->count[1];
  size_t off = 0;
  for (int q = 0; q < level; ++q) off += lane_tree_count(nleaf, q);
  const uint64_t c = lane_tree_count(nleaf, level);
  if (count) *count = c;
  if (top_level) *top_level = ix->qtop;
  if (capacity < c || (boxes == nullptr && cells == nullptr)) return PCLHIP_OK;
  std::vector<Box> tmp(c);
  for (int pass = 0; pass < 2; ++pass) {
    float* out = pass == 0 ? boxes : cells;
    if (out == nullptr) continue;
    if (pass == 1 && level == ix->qtop) {  // the root's cell is all of space (never stored, never read)
      for (int d = 0; d < 3; ++d) {
        out[d] = -INFINITY;
        out[3 + d] = INFINITY;
      }
      continue;
    }
    PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(tmp.data(), (pass == 0 ? ix->qbox : ix->qcell) + off, c * sizeof(Box),
                                         hipMemcpyDeviceToHost, ctx->stream));
    PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (uint64_t i = 0; i < c; ++i) {
      out[6 * i + 0] = tmp[i].lo.x; out[6 * i + 1] = tmp[i].lo.y; out[6 * i + 2] = tmp[i].lo.z;
      out[6 * i + 3] = tmp[i].hi.x; out[6 * i + 4] = tmp[i].hi.y; out[6 * i + 5] = tmp[i].hi.z;
    }
  }
  return PCLHIP_OK;
}

}  // extern "C"

// FEW queries against a LARGE index.  A wavefront owns 64 consecutive queries of the sorted batch and walks the tree for
// the box of all of them: compact when the batch is as dense as the cloud (a source cloud, self-queries), but 64 of a
// few thousand queries scattered over a 10M-point cloud span a sizeable part of it -- measured (round 6, random queries on
// the 10M-point sheet): 16 queries 117 ms, 1024 queries 34 ms, 16,384 queries 3.8 ms, 262,144 queries 0.9 ms per call.
// Such a batch is laid out with FEWER queries per wavefront: `fill` real queries in front of every 64-slot group, the
// other slots non-finite (they take no part in the search; their rows go to a dump row behind the results).  `fill` keeps
// the target points a group spans (about fill * n / nq) near SPARSE_SPAN.
constexpr uint64_t SPARSE_SPAN = 1024;
uint32_t pclhip::sparse_fill(uint64_t nq, uint64_t n_index) {
  if (n_index == 0 || nq == 0) return WAVE;
  const uint64_t f = SPARSE_SPAN * nq / n_index;
  if (f >= uint64_t(WAVE)) return WAVE;
  // a power of two: the sorted batch is in kd order (every aligned run of 16 * 4^j queries is a cell, and inside a run of
  // 16 the halves of the last binary cuts), so ALIGNED runs of 2^j queries are compact -- a run of 6 or 26 would straddle
  // cell boundaries, and the one that straddles the top-level cut spans the whole cloud (measured: 16,384 queries laid
  // out 6 per wavefront 13 ms, 65,536 queries 26 per wavefront 74 ms per call)
  uint32_t p = 1;
  while (uint64_t(p) * 2 <= f) p *= 2;
  return p;
}
__global__ void sparse_expand_kernel(const float4* __restrict__ q, uint32_t nq, uint32_t fill, float4* __restrict__ out,
                                     uint32_t n_out, uint32_t dump_row) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n_out) return;
  const uint32_t g = slot / WAVE, l = slot % WAVE, src = g * fill + l;
  const float qnan = __builtin_nanf("");
  out[slot] = (l < fill && src < nq) ? q[src] : make_float4(qnan, qnan, qnan, __uint_as_float(dump_row));
}
// one query per wavefront straight from the records (fill == 1: the order of the batch does not matter, so no ordering
// pass and none of its host waits): slot 64 g holds record g in the index's space, w = g; the other slots are padding
__global__ void sparse_load_one_per_group_kernel(const void* __restrict__ recs, size_t stride, uint32_t nq, float sx, float sy,
                                                 float sz, int scaled, float4* __restrict__ out) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= nq * uint32_t(WAVE)) return;
  const uint32_t g = slot / WAVE;
  const float qnan = __builtin_nanf("");
  float4 v = make_float4(qnan, qnan, qnan, __uint_as_float(nq));
  if (slot % WAVE == 0u) {
    const float* p = reinterpret_cast<const float*>(static_cast<const char*>(recs) + size_t(g) * stride);
    float x = p[0], y = p[1], z = p[2];
    if (scaled) {  // as kd_load_kernel maps the records (an axis with factor 0 does not exist in the representation)
      x = sx == 0.0f ? 0.0f : __fmul_rn(x, sx);
      y = sy == 0.0f ? 0.0f : __fmul_rn(y, sy);
      z = sz == 0.0f ? 0.0f : __fmul_rn(z, sz);
    }
    v = make_float4(x, y, z, __uint_as_float(g));
  }
  out[slot] = v;
}
// the sparse layout of a sorted batch (device memory of the context; *out == nullptr: the batch is dense enough as it is).
// Padding slots carry w = nq: a caller that scatters results by w keeps one dump row / slot behind its nq entries.
pclhip_status pclhip::sparse_layout(pclhip_ctx* ctx, const float4* q_sorted, uint64_t nq, uint64_t n_index, float4** out,
                                    uint32_t* n_out) {
  *out = nullptr;
  *n_out = uint32_t(nq);
  const uint32_t fill = sparse_fill(nq, n_index);
  if (fill > uint32_t(WAVE) / 2 || nq == 0) return PCLHIP_OK;
  const uint64_t ngroups = (nq + fill - 1) / fill;
  PCLHIP_REQUIRE(ctx, ngroups * WAVE < 0x7FFFFFFFull, "too many queries");
  float4* qe = nullptr;
  const uint32_t n_slots = uint32_t(ngroups * WAVE);
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &qe, size_t(n_slots) * sizeof(float4)));
  hipLaunchKernelGGL(sparse_expand_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, ctx->stream, q_sorted, uint32_t(nq), fill, qe,
                     n_slots, uint32_t(nq));
  if (hipGetLastError() != hipSuccess) {
    (void)dev_free(ctx, qe);
    set_error(ctx, "sparse layout: launch failed");
    return PCLHIP_ERR_HIP;
  }
  *out = qe;
  *n_out = n_slots;
  return PCLHIP_OK;
}

extern "C" {

// The per-point calls of PCL's search virtuals (Search::nearestKSearch(point, k, ...), one query at a time --
// impl/correspondence_estimation.hpp:163-175): at most one wavefront of host queries with host results goes through ONE
// pinned block -- the queries are written into it in the index's space, the kernel reads them and writes its rows there,
// one launch and one wait; no staging copy, no ordering pass (a single group needs none), no device allocation.
constexpr uint64_t KNN_FEW_QUERIES = 64;
constexpr int KNN_FEW_K = 32;
static pclhip_status knn_few(pclhip_index* ix, const void* queries, size_t stride, uint32_t nq, int k, int32_t* out_idx,
                             float* out_d2) {
  pclhip_ctx* ctx = ix->ctx;
  // worst case one query per wavefront (sparse_fill): 64 groups of 64 slots; rows of the queries + the dump row of the pads
  constexpr size_t Q_BYTES = KNN_FEW_QUERIES * WAVE * sizeof(float4), ROWS = (KNN_FEW_QUERIES + 1) * size_t(KNN_FEW_K);
  constexpr size_t BYTES = Q_BYTES + ROWS * (sizeof(int32_t) + sizeof(float));  // one size: the pinned cache always hits
  void* blk = nullptr;
  PCLHIP_CHECK_HIP(ctx, pinned_malloc(ctx, &blk, BYTES));
  float4* q = static_cast<float4*>(blk);
  int32_t* ri = reinterpret_cast<int32_t*>(static_cast<char*>(blk) + Q_BYTES);
  float* rd = reinterpret_cast<float*>(ri + ROWS);
  // (these queries are NOT sorted: against anything but a small index every query gets a wavefront of its own)
  const uint32_t fill = ix->n <= 65536u ? uint32_t(WAVE) : 1u;
  const uint32_t ngroups = (nq + fill - 1) / fill, n_slots = ngroups * uint32_t(WAVE);
  const float qnan = std::nanf("");
  float dump_w;
  std::memcpy(&dump_w, &nq, sizeof dump_w);
  for (uint32_t slot = 0; slot < n_slots; ++slot) {
    const uint32_t g = slot / uint32_t(WAVE), l = slot % uint32_t(WAVE), i = g * fill + l;
    if (l >= fill || i >= nq) {
      q[slot] = make_float4(qnan, qnan, qnan, dump_w);
      continue;
    }
    const float* p = reinterpret_cast<const float*>(static_cast<const char*>(queries) + size_t(i) * stride);
    float x = p[0], y = p[1], z = p[2];
    if (ix->scaled) {  // the index's (rescaled) space, as kd_load_kernel maps the records
      x = ix->scale[0] == 0.0f ? 0.0f : x * ix->scale[0];
      y = ix->scale[1] == 0.0f ? 0.0f : y * ix->scale[1];
      z = ix->scale[2] == 0.0f ? 0.0f : z * ix->scale[2];
    }
    float w;
    std::memcpy(&w, &i, sizeof(w));
    q[slot] = make_float4(x, y, z, w);
  }
  pclhip_status st = launch_knn(ix, q, n_slots, k, ri, rd);
  const hipError_t e = hipStreamSynchronize(ctx->stream);
  if (st == PCLHIP_OK && e == hipSuccess) {
    std::memcpy(out_idx, ri, size_t(nq) * size_t(k) * sizeof(int32_t));
    std::memcpy(out_d2, rd, size_t(nq) * size_t(k) * sizeof(float));
  }
  pinned_free(ctx, blk, BYTES);
  if (st != PCLHIP_OK) return st;
  PCLHIP_CHECK_HIP(ctx, e);
  return PCLHIP_OK;
}

pclhip_status pclhip_knn(pclhip_index* ix, const void* queries, size_t stride, uint64_t nq, int k, int32_t* out_idx,
                         float* out_d2) {
  if (!ix) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = ix->ctx;
  std::lock_guard<std::recursive_mutex> api_lock(ctx->api_mutex);  // safe under concurrent callers (pclhip.h)
  PCLHIP_REQUIRE(ctx, k >= 1, "k must be >= 1");
  PCLHIP_REQUIRE(ctx, stride >= 12 && stride % 4 == 0, "stride must be a multiple of 4 and >= 12 bytes");
  PCLHIP_REQUIRE(ctx, nq < 0x7FFFFFFFull, "too many queries");
  if (nq == 0) return PCLHIP_OK;
  PCLHIP_REQUIRE(ctx, queries && out_idx && out_d2, "null buffer");
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  if (nq <= KNN_FEW_QUERIES && k <= KNN_FEW_K && !is_device_pointer(queries) && !is_device_pointer(out_idx) &&
      !is_device_pointer(out_d2))
    return knn_few(ix, queries, stride, uint32_t(nq), k, out_idx, out_d2);
  DeviceGuard guard(ctx);
  const void* dq = nullptr;
  void* owned = nullptr;
  pclhip_status st = to_device(ctx, queries, size_t(nq) * stride, &dq, &owned);
  if (st != PCLHIP_OK) return st;
  guard.add(owned);
  const size_t cnt = size_t(nq) * size_t(k);
  const float4* q_run = nullptr;
  uint32_t nq_run = uint32_t(nq);
  bool sparse = false;
  if (sparse_fill(nq, ix->n) == 1u && nq * WAVE < 0x7FFFFFFFull) {
    // every query gets a wavefront of its own: the batch needs no order (and none of the ordering pass's host waits)
    float4* qe = nullptr;
    nq_run = uint32_t(nq * WAVE);
    PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &qe, size_t(nq_run) * sizeof(float4)));
    guard.add(qe);
    hipLaunchKernelGGL(sparse_load_one_per_group_kernel, dim3((nq_run + 255) / 256), dim3(256), 0, ctx->stream, dq, stride,
                       uint32_t(nq), ix->scale[0], ix->scale[1], ix->scale[2], ix->scaled ? 1 : 0, qe);
    PCLHIP_CHECK_HIP(ctx, hipGetLastError());
    q_run = qe;
    sparse = true;
  } else {
    float4* qs = nullptr;
    PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &qs, size_t(nq) * sizeof(float4)));
    guard.add(qs);
    uint32_t nf = 0;
    float lo[3], hi[3];
    st = spatial_order(ctx, dq, stride, nq, nullptr, 0, qs, uint32_t(nq), &nf, lo, hi, true, nullptr,
                       ix->scaled ? ix->scale : nullptr);  // queries live in the index's (rescaled) space
    if (st != PCLHIP_OK) return st;
    // a batch that is sparse against the index: fewer queries per wavefront (sparse_fill), results through buffers with a
    // dump row for the padding slots
    q_run = qs;
    float4* qe = nullptr;
    st = sparse_layout(ctx, qs, nq, ix->n, &qe, &nq_run);
    if (st != PCLHIP_OK) return st;
    sparse = qe != nullptr;
    if (sparse) {
      guard.add(qe);
      q_run = qe;
    }
  }
  int32_t* d_idx = out_idx;
  float* d_d2 = out_d2;
  const bool idx_dev = is_device_pointer(out_idx), d2_dev = is_device_pointer(out_d2);
  const size_t cnt_run = cnt + (sparse ? size_t(k) : 0);   // + the dump row
  if (!idx_dev || sparse) {
    PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &d_idx, cnt_run * sizeof(int32_t)));
    guard.add(d_idx);
  }
  if (!d2_dev || sparse) {
    PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &d_d2, cnt_run * sizeof(float)));
    guard.add(d_d2);
  }
  st = launch_knn(ix, q_run, nq_run, k, d_idx, d_d2);
  if (st != PCLHIP_OK) {
    (void)hipStreamSynchronize(ctx->stream);
    return st;
  }
  if (d_idx != out_idx)
    PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(out_idx, d_idx, cnt * sizeof(int32_t), idx_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                                         ctx->stream));
  if (d_d2 != out_d2)
    PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(out_d2, d_d2, cnt * sizeof(float), d2_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                                         ctx->stream));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PCLHIP_OK;
}

static pclhip_status normals_common(pclhip_index* ix, int k, double radius, const float viewpoint[3], void* out,
                                     size_t out_stride, uint64_t* out_nan_count) {
  if (!ix) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = ix->ctx;
  PCLHIP_REQUIRE(ctx, (k >= 1) != (radius > 0.0), "set either k >= 1 or a positive radius");
  PCLHIP_REQUIRE(ctx, !out || (out_stride >= 16 && out_stride % 4 == 0), "out stride must be >= 16 bytes");
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  const float zero[3] = {0, 0, 0};
  const float* vp = viewpoint ? viewpoint : zero;
  uint64_t nan = 0;
  pclhip_status st = (k >= 1) ? launch_normals(ix, k, vp, &nan) : launch_normals_radius(ix, radius, vp, &nan);
  if (st != PCLHIP_OK) return st;
  // points that were dropped from the index have NaN normals as well
  if (out_nan_count) *out_nan_count = nan + (ix->n_orig - ix->n);
  if (out && ix->n_orig > 0) {
    float4* dense = nullptr;
    PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &dense, size_t(ix->n_orig) * sizeof(float4)));
    DeviceGuard guard(ctx);
    guard.add(dense);
    hipLaunchKernelGGL(scatter_normals_kernel, dim3(unsigned((ix->n_orig + 255) / 256)), dim3(256), 0, ctx->stream,
                       ix->nrm, ix->rank, ix->n_orig, dense);
    PCLHIP_CHECK_HIP(ctx, hipGetLastError());
    st = copy_out_strided(ctx, out, out_stride, dense, sizeof(float4), ix->n_orig);
    if (st != PCLHIP_OK) return st;
  }
  return PCLHIP_OK;
}

pclhip_status pclhip_normals_records(pclhip_index* ix, int k, double radius, const float viewpoint[3], void* out,
                                     size_t record_bytes, size_t normal_offset, size_t curvature_offset,
                                     uint64_t* out_nan_count) {
  if (!ix) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = ix->ctx;
  PCLHIP_REQUIRE(ctx, out != nullptr, "null buffer");
  PCLHIP_REQUIRE(ctx, record_bytes % 4 == 0 && record_bytes >= 16 && record_bytes <= 256 && normal_offset % 4 == 0 &&
                          curvature_offset % 4 == 0 && normal_offset + 12 <= record_bytes && curvature_offset + 4 <= record_bytes &&
                          (curvature_offset + 4 <= normal_offset || curvature_offset >= normal_offset + 12),
                  "record layout: 4-byte aligned normal (12 bytes) and curvature (4 bytes) inside a record of 16..256 bytes");
  // the normals themselves (kept in the index), no output yet
  const pclhip_status st = normals_common(ix, k, radius, viewpoint, nullptr, 0, out_nan_count);
  if (st != PCLHIP_OK || ix->n_orig == 0) return st;
  const size_t bytes = size_t(ix->n_orig) * record_bytes;
  DeviceGuard guard(ctx);
  char* d = static_cast<char*>(out);
  const bool dev = is_device_pointer(out);
  if (!dev) {
    PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &d, bytes));
    guard.add(d);
  }
  hipLaunchKernelGGL(normal_records_kernel, dim3(unsigned((ix->n_orig + 255) / 256)), dim3(256), 0, ctx->stream, ix->nrm, ix->rank,
                     uint64_t(ix->n_orig), d, uint32_t(record_bytes), uint32_t(normal_offset), uint32_t(curvature_offset));
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  if (!dev) PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, ctx->stream));  // ONE linear copy
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PCLHIP_OK;
}

pclhip_status pclhip_normals(pclhip_index* ix, int k, const float viewpoint[3], void* out, size_t out_stride,
                             uint64_t* out_nan_count) {
  if (ix && k < 1) {
    set_error(ix->ctx, "k must be >= 1");
    return PCLHIP_ERR_INVALID;
  }
  return normals_common(ix, k, 0.0, viewpoint, out, out_stride, out_nan_count);
}

pclhip_status pclhip_normals_radius(pclhip_index* ix, double radius, const float viewpoint[3], void* out,
                                    size_t out_stride, uint64_t* out_nan_count) {
  if (ix && !(radius > 0.0)) {
    set_error(ix->ctx, "radius must be > 0");
    return PCLHIP_ERR_INVALID;
  }
  return normals_common(ix, 0, radius, viewpoint, out, out_stride, out_nan_count);
}

namespace {
// query j <- record sel[j] (or j) of a strided cloud, as a dense float4 with w = 1
__global__ void gather_queries_kernel(const void* pts, size_t stride, const int32_t* __restrict__ sel, uint64_t m,
                                      float4* __restrict__ out) {
  const uint64_t j = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
  if (j >= m) return;
  const uint64_t rec = sel ? uint64_t(sel[j]) : j;
  const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(pts) + rec * stride);
  out[j] = make_float4(p[0], p[1], p[2], 1.0f);
}
}  // namespace

pclhip_status pclhip_normals_at(pclhip_index* ix, const void* queries, size_t stride, uint64_t nq, const int32_t* indices,
                                uint64_t n_indices, int k, double radius, const float viewpoint[3], void* out,
                                size_t out_stride, uint64_t* out_nan_count) {
  if (!ix) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = ix->ctx;
  if (out_nan_count) *out_nan_count = 0;
  PCLHIP_REQUIRE(ctx, (k >= 1) != (radius > 0.0), "set either k >= 1 or a positive radius");
  PCLHIP_REQUIRE(ctx, stride >= 12 && stride % 4 == 0, "stride must be a multiple of 4 and >= 12 bytes");
  PCLHIP_REQUIRE(ctx, out_stride >= 16 && out_stride % 4 == 0, "out stride must be a multiple of 4 and >= 16 bytes");
  PCLHIP_REQUIRE(ctx, nq < 0x7FFFFFFFull && n_indices < 0x7FFFFFFFull, "too many queries");
  PCLHIP_REQUIRE(ctx, !ix->scaled, "normals need an index built in the cloud's own coordinates (no rescaling representation)");
  const uint64_t m = indices ? n_indices : nq;
  if (m == 0) return PCLHIP_OK;
  PCLHIP_REQUIRE(ctx, queries && out, "null buffer");
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  DeviceGuard guard(ctx);
  const void* dq = nullptr;
  void* owned = nullptr;
  pclhip_status st = to_device(ctx, queries, size_t(nq) * stride, &dq, &owned);
  if (st != PCLHIP_OK) return st;
  guard.add(owned);
  const void* dsel = nullptr;
  if (indices) {  // Feature::setIndices: untrusted, an index outside the cloud would be a wild device read
    st = to_device(ctx, indices, size_t(n_indices) * sizeof(int32_t), &dsel, &owned);
    if (st != PCLHIP_OK) return st;
    guard.add(owned);
    int* bad = nullptr;
    PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &bad, sizeof(int)));
    guard.add(bad);
    PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(check_indices_kernel, dim3(unsigned((n_indices + 255) / 256)), dim3(256), 0, ctx->stream,
                       static_cast<const int32_t*>(dsel), n_indices, nq, bad);
    int hbad = 0;
    PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PCLHIP_REQUIRE(ctx, hbad == 0, "indices must lie in [0, n)");
  }
  float4 *qd = nullptr, *dense = nullptr;
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &qd, size_t(m) * sizeof(float4)));
  guard.add(qd);
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &dense, size_t(m) * sizeof(float4)));
  guard.add(dense);
  hipLaunchKernelGGL(gather_queries_kernel, dim3(unsigned((m + 255) / 256)), dim3(256), 0, ctx->stream, dq, stride,
                     static_cast<const int32_t*>(dsel), m, qd);
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  const float zero[3] = {0, 0, 0};
  uint64_t nan = 0;
  if (ix->n == 0) {  // nothing to search: every normal is NaN (Feature::compute on an empty surface)
    PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(dense, 0xFF, size_t(m) * sizeof(float4), ctx->stream));
    nan = m;
  } else {
    st = launch_normals_at(ix, qd, uint32_t(m), k >= 1 ? k : 0, radius, viewpoint ? viewpoint : zero, dense, &nan);
    if (st != PCLHIP_OK) {
      (void)hipStreamSynchronize(ctx->stream);
      return st;
    }
  }
  if (out_nan_count) *out_nan_count = nan;
  return copy_out_strided(ctx, out, out_stride, dense, sizeof(float4), m);
}

namespace {
__global__ void scatter_cov_kernel(const double* __restrict__ cov_sorted, const uint32_t* __restrict__ rank, uint64_t n_orig,
                                   double* __restrict__ dense) {
  const uint64_t i = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
  if (i >= n_orig) return;
  const uint32_t pos = rank[i];
#pragma unroll
  for (int c = 0; c < 9; ++c) dense[i * 9 + c] = (pos == NO_INDEX) ? __builtin_nan("") : cov_sorted[size_t(pos) * 9 + c];
}
}  // namespace

pclhip_status pclhip_gicp_covariances(pclhip_index* ix, int k, double epsilon, double* out) {
  if (!ix || !out) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = ix->ctx;
  PCLHIP_REQUIRE(ctx, k >= 1 && k <= 32, "k_correspondences must be in 1..32 on this path (reference default 20)");
  // gicp.hpp:77-83: "Number or points in cloud is less than k_correspondences_"
  PCLHIP_REQUIRE(ctx, uint64_t(k) <= ix->n, "number of points in cloud is less than k_correspondences");
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  DeviceGuard guard(ctx);
  double* cov_sorted = nullptr;
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &cov_sorted, size_t(ix->n) * 9 * sizeof(double)));
  guard.add(cov_sorted);
  pclhip_status st = launch_gicp_covariances(ix, k, epsilon, cov_sorted);
  if (st != PCLHIP_OK) return st;
  const bool dev = is_device_pointer(out);
  double* dense = out;
  if (!dev) {
    PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &dense, size_t(ix->n_orig) * 9 * sizeof(double)));
    guard.add(dense);
  }
  hipLaunchKernelGGL(scatter_cov_kernel, dim3(unsigned((ix->n_orig + 255) / 256)), dim3(256), 0, ctx->stream, cov_sorted,
                     ix->rank, ix->n_orig, dense);
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  if (!dev)
    PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, dense, size_t(ix->n_orig) * 9 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PCLHIP_OK;
}

namespace {
// queries = the indexed points themselves; w = sorted position so that result row i belongs to sorted point i
__global__ void self_query_kernel(const float4* __restrict__ pts, uint32_t n, float4* __restrict__ q) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float4 p = pts[i];
    p.w = __uint_as_float(i);
    q[i] = p;
  }
}
// max over the points inside the box of their k-th neighbour distance; unsigned order == float order for d2 >= 0
__global__ void kth_max_kernel(const float4* __restrict__ pts, const float* __restrict__ d2, uint32_t n, int k, int has_box,
                               float lx, float ly, float lz, float hx, float hy, float hz, unsigned int* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  if (has_box && !(p.x >= lx && p.x <= hx && p.y >= ly && p.y <= hy && p.z >= lz && p.z <= hz)) return;
  const float v = d2[size_t(i) * k + (k - 1)];
  atomicMax(out, __float_as_uint(v));  // +inf (fewer than k points) dominates, as it should
}
}  // namespace

pclhip_status pclhip_index_kth_distance_max(pclhip_index* ix, int k, const float* box, double* out_max_d2) {
  if (!ix || !out_max_d2) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = ix->ctx;
  PCLHIP_REQUIRE(ctx, k >= 1, "k must be >= 1");
  *out_max_d2 = 0.0;
  if (ix->n == 0) return PCLHIP_OK;
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  DeviceScope scope(ctx);
  float4* q = nullptr;
  int32_t* idx = nullptr;
  float* d2 = nullptr;
  unsigned int* mx = nullptr;
  PCLHIP_CHECK_HIP(ctx, scope.alloc(&q, size_t(ix->n) * sizeof(float4)));
  PCLHIP_CHECK_HIP(ctx, scope.alloc(&idx, size_t(ix->n) * k * sizeof(int32_t)));
  PCLHIP_CHECK_HIP(ctx, scope.alloc(&d2, size_t(ix->n) * k * sizeof(float)));
  PCLHIP_CHECK_HIP(ctx, scope.alloc(&mx, sizeof(unsigned int)));
  PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(mx, 0, sizeof(unsigned int), ctx->stream));
  hipLaunchKernelGGL(self_query_kernel, dim3((ix->n + 255) / 256), dim3(256), 0, ctx->stream, ix->pts, ix->n, q);
  pclhip_status st = launch_knn(ix, q, ix->n, k, idx, d2);
  if (st != PCLHIP_OK) {
    (void)hipStreamSynchronize(ctx->stream);
    return st;
  }
  hipLaunchKernelGGL(kth_max_kernel, dim3((ix->n + 255) / 256), dim3(256), 0, ctx->stream, ix->pts, d2, ix->n, k, box ? 1 : 0,
                     box ? box[0] : 0.f, box ? box[1] : 0.f, box ? box[2] : 0.f, box ? box[3] : 0.f, box ? box[4] : 0.f,
                     box ? box[5] : 0.f, mx);
  unsigned int h = 0;
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(&h, mx, sizeof h, hipMemcpyDeviceToHost, ctx->stream);
  const hipError_t es = hipStreamSynchronize(ctx->stream);
  PCLHIP_CHECK_HIP(ctx, e);
  PCLHIP_CHECK_HIP(ctx, es);
  float f;
  std::memcpy(&f, &h, sizeof f);
  *out_max_d2 = double(f);
  return PCLHIP_OK;
}

pclhip_status pclhip_index_set_normals(pclhip_index* ix, const void* normals, size_t stride) {
  if (!ix) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = ix->ctx;
  PCLHIP_REQUIRE(ctx, normals != nullptr || ix->n_orig == 0, "null normals");
  PCLHIP_REQUIRE(ctx, stride >= 12 && stride % 4 == 0, "stride must be a multiple of 4 and >= 12 bytes");
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  DeviceGuard guard(ctx);
  const void* dn = nullptr;
  void* owned = nullptr;
  pclhip_status st = to_device(ctx, normals, strided_span(ix->n_orig, stride, 12), &dn, &owned);  // (see strided_span)
  if (st != PCLHIP_OK) return st;
  guard.add(owned);
  if (!ix->nrm) PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &ix->nrm, size_t(ix->n_pad) * sizeof(float4)));
  hipLaunchKernelGGL(gather_normals_kernel, dim3((ix->n_pad + 255) / 256), dim3(256), 0, ctx->stream, dn, stride,
                     ix->pts, ix->n, ix->n_pad, ix->nrm);
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ix->has_normals = true;
  return PCLHIP_OK;
}

// ------------------------------------------------------------------------------------------------
void pclhip_icp_params_default(pclhip_icp_params* p) {
  if (!p) return;
  p->max_iterations = 10;
  p->max_correspondence_distance = std::sqrt(DBL_MAX);
  p->transformation_epsilon = 0.0;
  p->transformation_rotation_epsilon = 0.0;
  p->euclidean_fitness_epsilon = -DBL_MAX;
  p->min_number_correspondences = 3;
  p->mode = PCLHIP_ICP_POINT_TO_POINT;
  p->failure_after_max_iterations = 0;
  p->max_iterations_similar_transforms = 0;
  p->mse_threshold_absolute = 1e-12;
}

pclhip_status pclhip_icp_create(pclhip_index* target, pclhip_icp** out) {
  if (!target || !out) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = target->ctx;
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  // The ICP kernels measure distances in the index's coordinates and accumulate the estimator in the cloud's:
  // they coincide only without a rescaling point representation.
  PCLHIP_REQUIRE(ctx, !target->scaled, "registration against an index built with rescale values is not supported");
  pclhip_icp* icp = new pclhip_icp();
  icp->ctx = ctx;
  icp->target = target;
  icp->prev_mse = DBL_MAX;
  if (dev_malloc(ctx, &icp->sums_dev, PCLHIP_ICP_NSUMS * sizeof(double)) != hipSuccess ||
      pinned_malloc(ctx, &icp->sums_host, PCLHIP_ICP_NSUMS * sizeof(double)) != hipSuccess ||
      // timing markers between kernels of one stream: device-scope release is enough (a system-scope
      // release would write the iteration's dirty lines back to memory at every marker)
      hipEventCreateWithFlags(&icp->ev0, hipEventReleaseToDevice) != hipSuccess ||
      hipEventCreateWithFlags(&icp->ev1, hipEventReleaseToDevice) != hipSuccess ||
      hipEventCreateWithFlags(&icp->ev_mid, hipEventReleaseToDevice) != hipSuccess) {
    set_error(ctx, "allocation failed in pclhip_icp_create");
    pclhip_icp_destroy(icp);
    return PCLHIP_ERR_HIP;
  }
  *out = icp;
  return PCLHIP_OK;
}

static void icp_free_source(pclhip_icp* icp) {
  if (icp->src_index) pclhip_index_destroy(icp->src_index);
  icp->src_index = nullptr;
  if (icp->src_records) (void)dev_free(icp->ctx, icp->src_records);
  icp->src_records = nullptr;
  icp->src_records_host = nullptr;
  icp->src_records_stride = 0;
  icp->src_records_n = 0;
  if (icp->own_block) (void)dev_free(icp->ctx, icp->own_block);
  icp->own_block = nullptr;
  icp->own_groups = 0;
  if (icp->lane_block) (void)dev_free(icp->ctx, icp->lane_block);
  icp->lane_block = nullptr;
  icp->lane_cap = 0;
  if (icp->src_sorted0) (void)dev_free(icp->ctx, icp->src_sorted0);
  if (icp->src_cur) (void)dev_free(icp->ctx, icp->src_cur);
  if (icp->src_nrm_sorted0) (void)dev_free(icp->ctx, icp->src_nrm_sorted0);
  if (icp->src_nrm_cur) (void)dev_free(icp->ctx, icp->src_nrm_cur);
  icp->src_nrm_sorted0 = icp->src_nrm_cur = nullptr;
  if (icp->match) (void)dev_free(icp->ctx, icp->match);
  if (icp->match_pos) (void)dev_free(icp->ctx, icp->match_pos);
  if (icp->keep) (void)dev_free(icp->ctx, icp->keep);
  icp->keep = nullptr;
  if (icp->match_d2) (void)dev_free(icp->ctx, icp->match_d2);
  if (icp->partials) (void)dev_free(icp->ctx, icp->partials);
  icp->src_sorted0 = icp->src_cur = nullptr;
  icp->match = nullptr;
  icp->match_pos = nullptr;
  icp->match_d2 = nullptr;
  icp->partials = nullptr;
}

void pclhip_icp_destroy(pclhip_icp* icp) {
  if (!icp) return;
  if (icp->ctx) {
    (void)hipSetDevice(icp->ctx->device);
    (void)hipStreamSynchronize(icp->ctx->stream);
  }
  icp_free_source(icp);
  if (icp->sums_dev) (void)dev_free(icp->ctx, icp->sums_dev);
  if (icp->sums_host) pinned_free(icp->ctx, icp->sums_host, PCLHIP_ICP_NSUMS * sizeof(double));
  if (icp->rej_state) (void)dev_free(icp->ctx, icp->rej_state);
  if (icp->rej_state_host) pinned_free(icp->ctx, icp->rej_state_host, sizeof(pclhip::RejState));
  if (icp->ctl) (void)dev_free(icp->ctx, icp->ctl);
  if (icp->ctl_host) pinned_free(icp->ctx, icp->ctl_host, sizeof(pclhip::IcpControl));
  if (icp->steps) pinned_free(icp->ctx, icp->steps, sizeof(pclhip::IcpStepRecord) * size_t(icp->steps_capacity));
  for (hipEvent_t e : icp->step_events)
    if (e) (void)hipEventDestroy(e);
  if (icp->ev0) (void)hipEventDestroy(icp->ev0);
  if (icp->ev1) (void)hipEventDestroy(icp->ev1);
  if (icp->ev_mid) (void)hipEventDestroy(icp->ev_mid);
  delete icp;
}

pclhip_status pclhip_icp_set_source(pclhip_icp* icp, const void* points, size_t stride, uint64_t n) {
  return pclhip_icp_set_source_indexed(icp, points, stride, n, nullptr, 0);
}

pclhip_status pclhip_icp_set_source_indexed(pclhip_icp* icp, const void* points, size_t stride, uint64_t n,
                                            const int32_t* indices, uint64_t n_indices) {
  if (!icp) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = icp->ctx;
  PCLHIP_REQUIRE(ctx, indices != nullptr || n_indices == 0, "null index buffer");
  PCLHIP_REQUIRE(ctx, stride >= 12 && stride % 4 == 0, "stride must be a multiple of 4 and >= 12 bytes");
  PCLHIP_REQUIRE(ctx, n < 0x7FFFFFFFull, "cloud too large for int32 indices");
  PCLHIP_REQUIRE(ctx, n == 0 || points != nullptr, "null point buffer");
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  dev_reserve_for_points(ctx, n);
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  icp_free_source(icp);
  DeviceGuard guard(ctx);
  const void* dp = nullptr;
  void* owned = nullptr;
  pclhip_status st = to_device(ctx, points, size_t(n) * stride, &dp, &owned);
  if (st != PCLHIP_OK) return st;
  if (owned != nullptr) {
    // a host cloud: its staged copy stays with the registration, so that the moved cloud an alignment hands back
    // (pclhip_icp_transform_source) does not upload the same records again.  The copy is the cloud AS IT WAS AT THIS CALL
    // (pclhip.h says so); it is not kept when it would hold more than an eighth of the device's memory.
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && size_t(n) * stride <= total_b / 8) {
      icp->src_records = owned;
      icp->src_records_host = points;
      icp->src_records_stride = stride;
      icp->src_records_n = n;
    } else {
      guard.add(owned);
    }
  }
  const void* dsel = nullptr;
  if (indices) {  // PCLBase::setIndices / setIndicesSource: only these points take part
    PCLHIP_REQUIRE(ctx, n_indices < 0x7FFFFFFFull, "too many indices");
    st = to_device(ctx, indices, size_t(n_indices) * sizeof(int32_t), &dsel, &owned);
    if (st != PCLHIP_OK) return st;
    guard.add(owned);
    if (n_indices > 0) {
      int* bad = nullptr;
      PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &bad, sizeof(int)));
      guard.add(bad);
      PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
      hipLaunchKernelGGL(check_indices_kernel, dim3(unsigned((n_indices + 255) / 256)), dim3(256), 0, ctx->stream,
                         static_cast<const int32_t*>(dsel), n_indices, n, bad);
      int hbad = 0;
      PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
      PCLHIP_REQUIRE(ctx, hbad == 0, "indices must lie in [0, n)");
    }
  }
  const uint64_t m = indices ? n_indices : n;
  icp->n_orig = n;
  icp->n = uint32_t(m);
  const size_t cap = m > 0 ? m : 1;
  icp->grid_blocks = icp_grid_blocks(ctx, icp->n);
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &icp->src_sorted0, cap * sizeof(float4)));
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &icp->src_cur, cap * sizeof(float4)));
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &icp->match, cap * sizeof(uint32_t)));
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &icp->match_pos, cap * sizeof(uint32_t)));
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &icp->match_d2, cap * sizeof(float)));
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &icp->partials, size_t(icp->grid_blocks) * PCLHIP_ICP_NSUMS * sizeof(double)));
  uint32_t nf = 0;
  float lo[3], hi[3];
  hipEvent_t e0 = nullptr, e1 = nullptr;
  DeviceScope timing(ctx);
  PCLHIP_CHECK_HIP(ctx, timing.event(&e0));
  PCLHIP_CHECK_HIP(ctx, timing.event(&e1));
  (void)hipEventRecord(e0, ctx->stream);
  st = spatial_order(ctx, dp, stride, n, static_cast<const int32_t*>(dsel), n_indices, icp->src_sorted0, uint32_t(m), &nf, lo,
                     hi, true, nullptr, nullptr);
  if (st != PCLHIP_OK) return st;
  (void)hipEventRecord(e1, ctx->stream);
  icp->n_finite = nf;
  for (int d = 0; d < 3; ++d) {  // the source's bounding box: how much of the target it covers (the search's group fill)
    icp->src_lo[d] = lo[d];
    icp->src_hi[d] = hi[d];
  }
  st = pclhip_icp_reset(icp);
  float ms = 0;
  if (st == PCLHIP_OK && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) icp->source_order_ms = ms;
  return st;
}

pclhip_status pclhip_icp_set_source_normals(pclhip_icp* icp, const void* normals, size_t stride) {
  if (!icp) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = icp->ctx;
  PCLHIP_REQUIRE(ctx, icp->src_sorted0 != nullptr, "set the source cloud first");
  PCLHIP_REQUIRE(ctx, normals != nullptr || icp->n == 0, "null normals");
  PCLHIP_REQUIRE(ctx, stride >= 12 && stride % 4 == 0, "stride must be a multiple of 4 and >= 12 bytes");
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  DeviceGuard guard(ctx);
  const void* dn = nullptr;
  void* owned = nullptr;
  pclhip_status st = to_device(ctx, normals, strided_span(icp->n_orig, stride, 12), &dn, &owned);  // (see strided_span)
  if (st != PCLHIP_OK) return st;
  guard.add(owned);
  const size_t cap = icp->n > 0 ? icp->n : 1;
  if (!icp->src_nrm_sorted0) PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &icp->src_nrm_sorted0, cap * sizeof(float4)));
  if (!icp->src_nrm_cur) PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &icp->src_nrm_cur, cap * sizeof(float4)));
  if (icp->n > 0) {
    hipLaunchKernelGGL(gather_normals_kernel, dim3((icp->n + 255) / 256), dim3(256), 0, ctx->stream, dn, stride,
                       icp->src_sorted0, icp->n, icp->n, icp->src_nrm_sorted0);
    PCLHIP_CHECK_HIP(ctx, hipGetLastError());
    PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(icp->src_nrm_cur, icp->src_nrm_sorted0, size_t(icp->n) * sizeof(float4),
                                         hipMemcpyDeviceToDevice, ctx->stream));
  }
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PCLHIP_OK;
}

pclhip_status pclhip_icp_set_enforce_same_direction_normals(pclhip_icp* icp, int enforce) {
  if (!icp) return PCLHIP_ERR_INVALID;
  icp->enforce_same_direction_normals = enforce != 0;
  return PCLHIP_OK;
}

pclhip_status pclhip_icp_set_allreduce(pclhip_icp* icp, pclhip_allreduce_fn fn, void* user) {
  if (!icp) return PCLHIP_ERR_INVALID;
  icp->allreduce = fn;
  icp->allreduce_user = user;
  return PCLHIP_OK;
}

pclhip_status pclhip_icp_set_rejectors(pclhip_icp* icp, const pclhip_rejector* list, int n) {
  if (!icp || n < 0 || (n > 0 && !list)) return PCLHIP_ERR_INVALID;
  for (int i = 0; i < n; ++i)
    PCLHIP_REQUIRE(icp->ctx, list[i].kind >= PCLHIP_REJ_DISTANCE && list[i].kind <= PCLHIP_REJ_TRIMMED, "unknown rejector kind");
  icp->rejectors.assign(list, list + n);
  return PCLHIP_OK;
}

double pclhip_icp_last_median_distance(const pclhip_icp* icp) {
  if (!icp || !icp->rej_state_host) return 0.0;
  (void)hipStreamSynchronize(icp->ctx->stream);  // the chain mirrors its state to pinned memory stream-ordered
  return icp->rej_state_host->median;
}

pclhip_status pclhip_icp_set_reciprocal(pclhip_icp* icp, int enable) {
  if (!icp) return PCLHIP_ERR_INVALID;
  icp->reciprocal = enable != 0;
  return PCLHIP_OK;
}

pclhip_status pclhip_icp_reset(pclhip_icp* icp) {
  if (!icp) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = icp->ctx;
  if (icp->n > 0) {
    PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(icp->src_cur, icp->src_sorted0, size_t(icp->n) * sizeof(float4),
                                         hipMemcpyDeviceToDevice, ctx->stream));
    if (icp->src_nrm_cur)
      PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(icp->src_nrm_cur, icp->src_nrm_sorted0, size_t(icp->n) * sizeof(float4),
                                           hipMemcpyDeviceToDevice, ctx->stream));
    // no seeds from a previous alignment: the first iteration searches from scratch
    PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(icp->match_pos, 0xFF, size_t(icp->n) * sizeof(uint32_t), ctx->stream));
    icp->seeds_cleared = true;
  }
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PCLHIP_OK;
}

pclhip_status pclhip_icp_iterate(pclhip_icp* icp, const float T_prev[16], double max_dist, int mode, double* sums) {
  if (!icp || !T_prev || !sums) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = icp->ctx;
  if (mode != PCLHIP_ICP_POINT_TO_POINT && mode != PCLHIP_ICP_POINT_TO_PLANE && mode != PCLHIP_ICP_SYMMETRIC) {
    set_error(ctx, "unknown ICP mode");
    return PCLHIP_ERR_INVALID;
  }
  if (mode == PCLHIP_ICP_SYMMETRIC && icp->src_nrm_cur == nullptr) {
    set_error(ctx, "the symmetric objective needs source normals (pclhip_icp_set_source_normals)");
    return PCLHIP_ERR_STATE;
  }
  if (mode != PCLHIP_ICP_POINT_TO_POINT && !icp->target->has_normals) {
    set_error(ctx, "point-to-plane ICP needs target normals (pclhip_normals / pclhip_index_set_normals)");
    return PCLHIP_ERR_STATE;
  }
  PCLHIP_REQUIRE(ctx, icp->src_cur != nullptr, "no source cloud set");
  {
    const pclhip_status sf = sharded_filters_ok(icp);
    if (sf != PCLHIP_OK) return sf;
  }
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  // correspondence_estimation.hpp:161,176: drop if double(d2) > max_dist*max_dist
  const double md2 = max_dist * max_dist;
  const bool use_max = md2 < double(FLT_MAX);
  const float fmax2 = use_max ? float_at_most(md2) : FLT_MAX;
  pclhip_status st = launch_icp_iterate(icp, T_prev, fmax2, use_max, mode);  // incl. the all-reduce of the record
  if (st != PCLHIP_OK) return st;
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(icp->sums_host, icp->sums_dev, PCLHIP_ICP_NSUMS * sizeof(double),
                                       hipMemcpyDeviceToHost, ctx->stream));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::memcpy(sums, icp->sums_host, PCLHIP_ICP_NSUMS * sizeof(double));
  float ms = 0;
  if (icp->n > 0 && hipEventElapsedTime(&ms, icp->ev0, icp->ev1) == hipSuccess) icp->last_kernel_ms = ms;
  icp->last_search_ms = icp->last_kernel_ms;
  if (icp->n > 0 && icp->mid_recorded && hipEventElapsedTime(&ms, icp->ev0, icp->ev_mid) == hipSuccess)
    icp->last_search_ms = ms;
  return PCLHIP_OK;
}

double pclhip_icp_last_kernel_ms(const pclhip_icp* icp) { return icp ? icp->last_kernel_ms : 0.0; }
double pclhip_icp_source_order_ms(const pclhip_icp* icp) { return icp ? icp->source_order_ms : 0.0; }
double pclhip_icp_last_search_ms(const pclhip_icp* icp) { return icp ? icp->last_search_ms : 0.0; }
double pclhip_index_last_kernel_ms(const pclhip_index* ix) { return ix ? ix->last_kernel_ms : 0.0; }

pclhip_status pclhip_solve_transformation(const double* sums, int mode, float* T) {
  if (!sums || !T) return PCLHIP_ERR_INVALID;
  if (mode == PCLHIP_ICP_POINT_TO_PLANE)
    solve_point_to_plane(sums, T);
  else if (mode == PCLHIP_ICP_SYMMETRIC)
    solve_symmetric(sums, T);
  else if (mode == PCLHIP_ICP_POINT_TO_POINT)
    solve_point_to_point(sums, T);
  else
    return PCLHIP_ERR_INVALID;
  return PCLHIP_OK;
}

// IterativeClosestPoint::computeTransformation, registration/include/pcl/registration/impl/icp.hpp:113-268
// with DefaultConvergenceCriteria::hasConverged, impl/default_convergence_criteria.hpp:49-140.
pclhip_status pclhip_icp_align(pclhip_icp* icp, const pclhip_icp_params* params, const float* guess,
                               pclhip_icp_result* res) {
  if (!icp || !params || !res) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = icp->ctx;
  std::memset(res, 0, sizeof *res);
  // The loop runs on the device (icp_loop.hip): no read-back, host solve or reset copy between iterations; a
  // rejector chain and reciprocal correspondences are part of it (counts, ranks and thresholds stay in device memory; the
  // source index of the reciprocal test is built once and refitted per iteration: rejectors.hip).  A host-driven loop
  // is what a caller composes from pclhip_icp_iterate + pclhip_solve_transformation + pclhip_convergence_has_converged
  // (tests/test_gpu_loop.py does, as the twin of this one).
  {
    const pclhip_status sf = sharded_filters_ok(icp);
    if (sf != PCLHIP_OK) return sf;
  }
  if (params->mode != PCLHIP_ICP_POINT_TO_POINT && params->mode != PCLHIP_ICP_POINT_TO_PLANE &&
      params->mode != PCLHIP_ICP_SYMMETRIC) {
    set_error(ctx, "unknown ICP mode");
    return PCLHIP_ERR_INVALID;
  }
  if (params->mode == PCLHIP_ICP_SYMMETRIC && icp->src_nrm_cur == nullptr) {
    set_error(ctx, "the symmetric objective needs source normals (pclhip_icp_set_source_normals)");
    return PCLHIP_ERR_STATE;
  }
  if (params->mode != PCLHIP_ICP_POINT_TO_POINT && !icp->target->has_normals) {
    set_error(ctx, "point-to-plane ICP needs target normals (pclhip_normals / pclhip_index_set_normals)");
    return PCLHIP_ERR_STATE;
  }
  PCLHIP_REQUIRE(ctx, icp->src_cur != nullptr, "no source cloud set");
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  return icp_align_device(icp, params, guess, res);
}

namespace {
__global__ void pack_float4_kernel(const void* in, size_t stride, uint32_t n, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(in) + size_t(i) * stride);
  out[i] = make_float4(p[0], p[1], p[2], 0.0f);
}
}  // namespace

static pclhip_status estimate_pairs_common(pclhip_ctx* ctx, int mode, const void* src, size_t src_stride,
                                           const void* src_normals, size_t src_normals_stride, const void* tgt,
                                           size_t tgt_stride, const void* tgt_normals, size_t tgt_normals_stride,
                                           const float* weights, uint64_t n, int enforce, float T[16], double* sums_out) {
  if (!ctx || !T) return PCLHIP_ERR_INVALID;
  PCLHIP_REQUIRE(ctx, mode >= PCLHIP_ICP_POINT_TO_POINT && mode <= PCLHIP_ICP_SYMMETRIC, "unknown estimator");
  PCLHIP_REQUIRE(ctx, n < 0x7FFFFFFFull, "too many pairs");
  PCLHIP_REQUIRE(ctx, n == 0 || (src && tgt), "null point buffer");
  PCLHIP_REQUIRE(ctx, mode == PCLHIP_ICP_POINT_TO_POINT || n == 0 || tgt_normals, "this estimator needs target normals");
  PCLHIP_REQUIRE(ctx, mode != PCLHIP_ICP_SYMMETRIC || n == 0 || src_normals, "this estimator needs source normals");
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  double sums[PCLHIP_ICP_NSUMS];
  std::memset(sums, 0, sizeof sums);
  if (n > 0) {
    DeviceGuard guard(ctx);
    const void* in[4] = {src, mode == PCLHIP_ICP_SYMMETRIC ? src_normals : nullptr, tgt,
                         mode != PCLHIP_ICP_POINT_TO_POINT ? tgt_normals : nullptr};
    const size_t strides[4] = {src_stride, src_normals_stride, tgt_stride, tgt_normals_stride};
    float4* packed[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int a = 0; a < 4; ++a) {
      if (!in[a]) continue;
      PCLHIP_REQUIRE(ctx, strides[a] >= 12 && strides[a] % 4 == 0, "stride must be a multiple of 4 and >= 12 bytes");
      const void* d = nullptr;
      void* owned = nullptr;
      // exactly the bytes the kernel reads: three floats of every record.  The pointer may sit INSIDE the caller's records
      // (the normals of a PointNormal array: base + 16, stride 48) -- n * stride from there runs 16 bytes past the array
      pclhip_status st = to_device(ctx, in[a], strided_span(n, strides[a], 12), &d, &owned);
      if (st != PCLHIP_OK) return st;
      guard.add(owned);
      void* buf = nullptr;
      PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &buf, size_t(n) * sizeof(float4)));
      guard.add(buf);
      packed[a] = static_cast<float4*>(buf);
      hipLaunchKernelGGL(pack_float4_kernel, dim3((uint32_t(n) + 255) / 256), dim3(256), 0, ctx->stream, d, strides[a],
                         uint32_t(n), packed[a]);
    }
    const float* dw = nullptr;
    if (weights) {
      const void* d = nullptr;
      void* owned = nullptr;
      pclhip_status stw = to_device(ctx, weights, size_t(n) * sizeof(float), &d, &owned);
      if (stw != PCLHIP_OK) return stw;
      guard.add(owned);
      dw = static_cast<const float*>(d);
    }
    PCLHIP_CHECK_HIP(ctx, hipGetLastError());
    pclhip_status st = launch_estimate_pairs(ctx, mode, packed[0], packed[1], packed[2], packed[3], dw, uint32_t(n),
                                             enforce != 0, sums);
    if (st != PCLHIP_OK) return st;
  }
  if (sums_out) std::memcpy(sums_out, sums, sizeof sums);
  return pclhip_solve_transformation(sums, mode, T);
}

pclhip_status pclhip_estimate_rigid_transformation(pclhip_ctx* ctx, int mode, const void* src, size_t src_stride,
                                                   const void* src_normals, size_t src_normals_stride, const void* tgt,
                                                   size_t tgt_stride, const void* tgt_normals, size_t tgt_normals_stride,
                                                   uint64_t n, int enforce, float T[16], double* sums_out) {
  return estimate_pairs_common(ctx, mode, src, src_stride, src_normals, src_normals_stride, tgt, tgt_stride, tgt_normals,
                               tgt_normals_stride, nullptr, n, enforce, T, sums_out);
}

pclhip_status pclhip_estimate_rigid_transformation_weighted(pclhip_ctx* ctx, const void* src, size_t src_stride,
                                                            const void* tgt, size_t tgt_stride, const void* tgt_normals,
                                                            size_t tgt_normals_stride, const float* weights, uint64_t n,
                                                            float T[16], double* sums_out) {
  if (!ctx) return PCLHIP_ERR_INVALID;
  PCLHIP_REQUIRE(ctx, n == 0 || weights != nullptr, "null weights");
  return estimate_pairs_common(ctx, PCLHIP_ICP_POINT_TO_PLANE, src, src_stride, nullptr, 0, tgt, tgt_stride, tgt_normals,
                               tgt_normals_stride, weights, n, 1, T, sums_out);
}

pclhip_status pclhip_icp_fitness_score(pclhip_icp* icp, const float T[16], double max_range, double* score,
                                       uint64_t* nr) {
  if (!icp || !T || !score) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = icp->ctx;
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  PCLHIP_REQUIRE(ctx, icp->target != nullptr, "no target index");
  uint64_t n_used = 0;
  pclhip_status st = launch_fitness_score(icp, T, max_range, score, &n_used);
  if (nr) *nr = n_used;
  return st;
}

pclhip_status pclhip_icp_fetch_correspondence_records(pclhip_icp* icp, void* out, uint64_t capacity, uint64_t* out_n) {
  if (!icp || !out_n) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = icp->ctx;
  *out_n = 0;
  if (icp->n == 0) return PCLHIP_OK;
  const bool filtered = icp->reciprocal || !icp->rejectors.empty();
  for (const pclhip_rejector& r : icp->rejectors) {
    if (r.kind == PCLHIP_REJ_ONE_TO_ONE || r.kind == PCLHIP_REJ_TRIMMED) {
      set_error(ctx, "correspondence records come in query order: OneToOne / Trimmed re-order the list (pclhip_icp_fetch_correspondences)");
      return PCLHIP_ERR_STATE;
    }
  }
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  DeviceGuard guard(ctx);
  const size_t no = size_t(icp->n_orig);  // dense arrays over the ORIGINAL source records (a subset leaves gaps)
  int32_t* dm = nullptr;
  float* dd = nullptr;
  uint32_t *flag = nullptr, *pos = nullptr, *tot = nullptr;
  uint2* partial = nullptr;
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &dm, no * sizeof(int32_t)));
  guard.add(dm);
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &dd, no * sizeof(float)));
  guard.add(dd);
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &flag, no * sizeof(uint32_t)));
  guard.add(flag);
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &pos, no * sizeof(uint32_t)));
  guard.add(pos);
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &partial, ((no + SC_BLOCK - 1) / SC_BLOCK) * sizeof(uint2)));
  guard.add(partial);
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &tot, 4 * sizeof(uint32_t)));
  guard.add(tot);
  hipStream_t s = ctx->stream;
  PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(dm, 0xFF, no * sizeof(int32_t), s));
  PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(tot, 0, 4 * sizeof(uint32_t), s));
  hipLaunchKernelGGL(scatter_matches_kernel, dim3((icp->n + 255) / 256), dim3(256), 0, s, icp->src_cur, icp->match, icp->match_d2,
                     filtered ? icp->keep : nullptr, icp->n, dm, dd);
  const unsigned blocks = unsigned((no + 255) / 256);
  hipLaunchKernelGGL(match_flag_kernel, dim3(blocks), dim3(256), 0, s, dm, uint64_t(no), flag);
  launch_scan_u32(s, flag, no, partial, tot, pos);
  uint32_t total = 0;
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(&total, tot, sizeof total, hipMemcpyDeviceToHost, s));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
  *out_n = total;
  if (total == 0) return PCLHIP_OK;
  if (!out || capacity < total) {
    set_error(ctx, "correspondence records: output capacity too small (out_n holds the required number)");
    return PCLHIP_ERR_OVERFLOW;
  }
  CorrRecord* drec = static_cast<CorrRecord*>(out);
  const bool dev = is_device_pointer(out);
  if (!dev) {
    PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &drec, size_t(total) * sizeof(CorrRecord)));
    guard.add(drec);
  }
  hipLaunchKernelGGL(match_emit_kernel, dim3(blocks), dim3(256), 0, s, dm, dd, pos, uint64_t(no), drec);
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  if (!dev) PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, drec, size_t(total) * sizeof(CorrRecord), hipMemcpyDeviceToHost, s));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(s));
  return PCLHIP_OK;
}

pclhip_status pclhip_icp_fetch_correspondences(pclhip_icp* icp, int32_t* index_query, int32_t* index_match,
                                               float* distance, uint64_t* out_n) {
  if (!icp || !out_n) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = icp->ctx;
  *out_n = 0;
  if (icp->n == 0) return PCLHIP_OK;
  PCLHIP_REQUIRE(ctx, index_query && index_match && distance, "null buffer");
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  int32_t* dm = nullptr;
  float* dd = nullptr;
  DeviceGuard guard(ctx);
  const size_t no = size_t(icp->n_orig);  // dense arrays over the ORIGINAL source records (a subset leaves gaps)
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &dm, no * sizeof(int32_t)));
  guard.add(dm);
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &dd, no * sizeof(float)));
  guard.add(dd);
  PCLHIP_CHECK_HIP(ctx, hipMemsetAsync(dm, 0xFF, no * sizeof(int32_t), ctx->stream));
  const bool filtered = icp->reciprocal || !icp->rejectors.empty();
  hipLaunchKernelGGL(scatter_matches_kernel, dim3((icp->n + 255) / 256), dim3(256), 0, ctx->stream, icp->src_cur,
                     icp->match, icp->match_d2, filtered ? icp->keep : nullptr, icp->n, dm, dd);
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  std::vector<int32_t> hm(no);
  std::vector<float> hd(no);
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(hm.data(), dm, no * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(hd.data(), dd, no * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<int32_t> q, m;
  std::vector<float> d;
  q.reserve(icp->n);
  m.reserve(icp->n);
  d.reserve(icp->n);
  for (size_t i = 0; i < no; ++i)
    if (hm[i] >= 0) {
      q.push_back(int32_t(i));
      m.push_back(hm[i]);
      d.push_back(hd[i]);
    }
  const size_t c = q.size();
  if (filtered && icp->trim_pending && icp->rej_state_host && icp->rej_state_host->trimmed) icp->fetch_order = 2;
  if (filtered && icp->fetch_order != 0 && c > 1) {
    // the reference's output order after the chain: ONE_TO_ONE sorts by (match, distance)
    // (correspondence_rejection_one_to_one.cpp:49-51), TRIMMED by distance (..._trimmed.cpp:53-56)
    std::vector<size_t> ord(c);
    for (size_t i = 0; i < c; ++i) ord[i] = i;
    if (icp->fetch_order == 1)
      std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) {
        return m[a] < m[b] || (m[a] == m[b] && d[a] < d[b]);
      });
    else
      std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return d[a] < d[b]; });
    std::vector<int32_t> q2(c), m2(c);
    std::vector<float> d2v(c);
    for (size_t i = 0; i < c; ++i) {
      q2[i] = q[ord[i]];
      m2[i] = m[ord[i]];
      d2v[i] = d[ord[i]];
    }
    q.swap(q2);
    m.swap(m2);
    d.swap(d2v);
  }
  auto put = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
    if (bytes == 0) return hipSuccess;
    if (is_device_pointer(dst)) return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    std::memcpy(dst, src, bytes);
    return hipSuccess;
  };
  PCLHIP_CHECK_HIP(ctx, put(index_query, q.data(), c * sizeof(int32_t)));
  PCLHIP_CHECK_HIP(ctx, put(index_match, m.data(), c * sizeof(int32_t)));
  PCLHIP_CHECK_HIP(ctx, put(distance, d.data(), c * sizeof(float)));
  *out_n = c;
  return PCLHIP_OK;
}

pclhip_status pclhip_transform_cloud(pclhip_ctx* ctx, const float* T, int order, const void* in, void* out,
                                     size_t stride, uint64_t n, size_t normals_offset_bytes) {
  return transform_cloud_impl(ctx, T, order, in, nullptr, out, stride, n, normals_offset_bytes);
}

pclhip_status pclhip_icp_transform_source(pclhip_icp* icp, const float* T, int order, const void* in, void* out,
                                          size_t stride, uint64_t n, size_t normals_offset_bytes) {
  if (!icp) return PCLHIP_ERR_INVALID;
  const bool resident = icp->src_records != nullptr && in == icp->src_records_host && stride == icp->src_records_stride &&
                        n == icp->src_records_n;
  return transform_cloud_impl(icp->ctx, T, order, in, resident ? icp->src_records : nullptr, out, stride, n,
                              normals_offset_bytes);
}

// `resident`: a device copy of the records at `in` that is known to be current (or nullptr: `in` is staged)
static pclhip_status transform_cloud_impl(pclhip_ctx* ctx, const float* T, int order, const void* in, const void* resident,
                                          void* out, size_t stride, uint64_t n, size_t normals_offset_bytes) {
  if (!ctx || !T) return PCLHIP_ERR_INVALID;
  PCLHIP_REQUIRE(ctx, stride >= 12 && stride % 4 == 0, "stride must be a multiple of 4 and >= 12 bytes");
  PCLHIP_REQUIRE(ctx, normals_offset_bytes == 0 || normals_offset_bytes + 12 <= stride, "bad normals offset");
  if (n == 0) return PCLHIP_OK;
  PCLHIP_REQUIRE(ctx, in && out, "null buffer");
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  DeviceGuard guard(ctx);
  const size_t bytes = size_t(n) * stride;
  const void* din = resident;
  void* owned = nullptr;
  pclhip_status st = PCLHIP_OK;
  if (din == nullptr) st = to_device(ctx, in, bytes, &din, &owned);
  if (st != PCLHIP_OK) return st;
  guard.add(owned);
  void* dout = out;
  const bool out_dev = is_device_pointer(out);
  if (!out_dev) {
    if (owned && in == out) {
      dout = owned;  // in-place on the staged copy
    } else {
      PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &dout, bytes));
      guard.add(dout);
      // other bytes of each record pass through unchanged
      PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(dout, din, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    }
  } else if (dout != din) {
    PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(dout, din, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  }
  Mat44 M;
  std::memcpy(M.m, T, sizeof M.m);
  hipLaunchKernelGGL(transform_cloud_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, ctx->stream, M, order, din,
                     dout, stride, n, normals_offset_bytes);
  PCLHIP_CHECK_HIP(ctx, hipGetLastError());
  if (!out_dev) PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, ctx->stream));
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PCLHIP_OK;
}

}  // extern "C"
