// wavesim.hpp -- TEST INFRASTRUCTURE, never part of the product: a lane-accurate CPU emulation of the wavefront
// execution model, so that the kernels of pcl_amd/csrc (the very same sources, compiled for the host with
// -DPCLHIP_WAVESIM and this header force-included) can be run against the oracle in the CPU test tier, where no GPU
// exists.  tests/wavesim/Makefile builds tests/wavesim/libpclhip_wavesim.so from them; only tests/test_wavesim.py loads
// it (through PCLHIP_LIB, in a subprocess).  pcl_amd/_lib.py never picks it up by itself, bench.py never sees it, and
// nothing is ever timed on it: it answers "does this kernel compute what the oracle computes", not "how fast".
//
// Model.  Every lane of a workgroup is a fiber (its own stack, cooperative switches, all fibers of a workgroup on one OS
// thread); workgroups of a launch are spread over a pool of OS threads.  A lane runs until it reaches a CROSS-LANE
// operation -- ballot, readlane, readfirstlane, shuffles, DPP moves, wave_barrier, __syncthreads -- where it publishes its
// operand and yields; when all live lanes of the wavefront (workgroup, for __syncthreads) have arrived at the SAME call
// site they are released and read each other's operands.  That is lock step at the granularity the hardware guarantees
// anything about: between two such operations the lanes of a wavefront exchange nothing (LDS communication inside a
// wavefront needs a wave_barrier in the sources anyway -- the compiler may otherwise reorder it).  Lanes that arrive at
// DIFFERENT call sites are a divergence the kernels are not supposed to have around cross-lane operations: the run stops
// with both source locations.  __shared__ is `static thread_local` (one copy per OS thread = per running workgroup),
// global_load_lds copies 16 bytes per active lane to (uniform LDS base + lane * 16), atomics are the host's.
// Arithmetic: the same float operations in the same order (-ffp-contract=off; __fmaf_rn = fmaf; sqrt and division are
// correctly rounded as on the device; v_rsq / v_sqrt approximations appear only inside bounds that carry allowances).
#pragma once
#ifndef PCLHIP_WAVESIM
#error "wavesim.hpp is for -DPCLHIP_WAVESIM host builds of the kernels (tests/wavesim/Makefile)"
#endif

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <sched.h>
#include <cstring>
#include <functional>

namespace wavesim {

struct Idx3 {
  unsigned x, y, z;
};
struct LaneCtx {  // what a fiber sees of itself
  Idx3 tid, bid, bdim, gdim;
};
extern thread_local LaneCtx* cur;  // the fiber running on this OS thread

// ---- cross-lane plumbing (wavesim_rt.cpp) ----------------------------------------------------------------------------
// publish `v`, wait for the wavefront at call site `site`, return the table of the 64 lanes' values (valid until the
// lane's next cross-lane operation) and the mask of live lanes
const uint64_t* exchange(uint64_t v, const char* site, uint64_t* live_mask);
void block_barrier(const char* site);
// A divergent branch that holds cross-lane operations: its lanes run as a partial wavefront while the others wait where
// the branch rejoins (the hardware's execution mask).  The sources mark such branches with PCLHIP_LANE_MASKED_REGION.
void masked_region_enter();
void masked_region_leave();
struct MaskedRegion {
  MaskedRegion() { masked_region_enter(); }
  ~MaskedRegion() { masked_region_leave(); }
};
void launch(dim3 grid, dim3 block, const std::function<void()>& body);

#define WAVESIM_STR2(x) #x
#define WAVESIM_STR(x) WAVESIM_STR2(x)
#define WAVESIM_SITE __FILE__ ":" WAVESIM_STR(__LINE__)

template <class T>
inline uint64_t to_bits(T v) {
  static_assert(sizeof(T) <= 8, "cross-lane operands are at most 64 bits");
  uint64_t b = 0;
  std::memcpy(&b, &v, sizeof(T));
  return b;
}
template <class T>
inline T from_bits(uint64_t b) {
  T v;
  std::memcpy(&v, &b, sizeof(T));
  return v;
}
inline unsigned lane_id() { return cur->tid.x & 63u; }

inline uint64_t ballot(bool p, const char* site) {
  uint64_t live;
  const uint64_t* t = exchange(p ? 1u : 0u, site, &live);
  uint64_t m = 0;
  for (int l = 0; l < 64; ++l)
    if (((live >> l) & 1u) && t[l]) m |= 1ull << l;
  return m;
}
template <class T>
inline T read_lane(T v, int lane, const char* site) {
  uint64_t live;
  const uint64_t* t = exchange(to_bits(v), site, &live);
  return from_bits<T>(t[lane & 63]);
}
template <class T>
inline T read_first(T v, const char* site) {
  uint64_t live;
  const uint64_t* t = exchange(to_bits(v), site, &live);
  return from_bits<T>(t[live ? __builtin_ctzll(live) : 0]);
}
inline void wave_sync(const char* site) {
  uint64_t live;
  (void)exchange(0, site, &live);
}
// DPP controls the sources use: quad_perm (0x00-0xFF), row_half_mirror 0x141, row_mirror 0x140
inline int dpp_source_lane(int lane, int ctrl) {
  if (ctrl >= 0 && ctrl <= 0xFF) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
  if (ctrl == 0x141) return (lane & ~7) | (7 - (lane & 7));
  if (ctrl == 0x140) return (lane & ~15) | (15 - (lane & 15));
  __builtin_trap();
}
inline int update_dpp(int /*old*/, int src, int ctrl, int /*row_mask*/, int /*bank_mask*/, bool /*bound_ctrl*/, const char* site) {
  uint64_t live;
  const uint64_t* t = exchange(to_bits(src), site, &live);
  return from_bits<int>(t[dpp_source_lane(int(lane_id()), ctrl)]);
}
// global_load_lds_dwordx4: every ACTIVE lane moves `size` bytes from its own global address to the wave-uniform LDS
// base + lane * size (+ offset).  The copy is immediate; s_waitcnt is a no-op here.
inline void global_load_lds(const void* g, void* lds_base, unsigned size, unsigned offset, unsigned /*aux*/) {
  std::memcpy(static_cast<char*>(lds_base) + offset + size_t(lane_id()) * size, g, size);
}

}  // namespace wavesim

// ---- the language -------------------------------------------------------------------------------------------------------
#undef __shared__
#define __shared__ static thread_local
#ifndef __launch_bounds__
#define __launch_bounds__(...)
#endif
#undef __forceinline__
#define __forceinline__ inline __attribute__((always_inline))
#define threadIdx (::wavesim::cur->tid)
#define blockIdx (::wavesim::cur->bid)
#define blockDim (::wavesim::cur->bdim)
#define gridDim (::wavesim::cur->gdim)
#define warpSize 64

#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  ::wavesim::launch(dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); })

#define PCLHIP_LANE_MASKED_REGION ::wavesim::MaskedRegion pclhip_lane_masked_region_guard
#define __syncthreads() ::wavesim::block_barrier(WAVESIM_SITE)
#define __builtin_amdgcn_ballot_w64(p) ::wavesim::ballot((p), WAVESIM_SITE)
#define __builtin_amdgcn_readlane(v, l) ::wavesim::read_lane<int>((v), (l), WAVESIM_SITE)
#define __builtin_amdgcn_readfirstlane(v) ::wavesim::read_first<int>((v), WAVESIM_SITE)
#define __builtin_amdgcn_wave_barrier() ::wavesim::wave_sync(WAVESIM_SITE)
#define __builtin_amdgcn_update_dpp(o, s, c, rm, bm, bc) ::wavesim::update_dpp((o), (s), (c), (rm), (bm), (bc), WAVESIM_SITE)
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) \
  ::wavesim::global_load_lds((const void*)(g), (void*)(l), (size), (off), (aux))
#define __builtin_amdgcn_alignbit(hi, lo, sh) \
  uint32_t(((uint64_t(uint32_t(hi)) << 32) | uint64_t(uint32_t(lo))) >> ((sh) & 31))
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)
// a polling lane gives its OS thread away (the workgroup it waits for may share the core)
#define __builtin_amdgcn_s_sleep(n) ::sched_yield()
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_AGENT 4   // clang's value; the host build of __hip_atomic_* ignores the scope
#endif
#define __builtin_amdgcn_fmed3f(a, b, c) ::wavesim_fmed3(a, b, c)
inline float wavesim_fmed3(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }

// shuffles (width 64)
#define __shfl(v, src) ::wavesim::read_lane((v), int(src), WAVESIM_SITE)
#define __shfl_xor(v, mask) ::wavesim::read_lane((v), int(::wavesim::lane_id() ^ unsigned(mask)), WAVESIM_SITE)
#define __shfl_up(v, delta) \
  ::wavesim::read_lane((v), (::wavesim::lane_id() >= unsigned(delta)) ? int(::wavesim::lane_id() - unsigned(delta)) : int(::wavesim::lane_id()), WAVESIM_SITE)

// ---- arithmetic intrinsics: the same IEEE operations ----------------------------------------------------------------
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline float __fdiv_rn(float a, float b) { return a / b; }
inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
inline float __fsqrt_rn(float a) { return sqrtf(a); }
inline float __frsqrt_rn(float a) { return 1.0f / sqrtf(a); }
inline uint32_t __float_as_uint(float f) { return wavesim::from_bits<uint32_t>(wavesim::to_bits(f)); }
inline int __float_as_int(float f) { return wavesim::from_bits<int>(wavesim::to_bits(f)); }
inline float __uint_as_float(uint32_t u) { return wavesim::from_bits<float>(uint64_t(u)); }
inline float __int_as_float(int i) { return wavesim::from_bits<float>(uint64_t(uint32_t(i))); }
inline long long clock64() {
  static thread_local long long t = 0;
  return t += 7;
}
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

template <class T>
inline T atomicAdd(T* p, T v) {
  return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
inline float atomicAdd(float* p, float v) {
  uint32_t* u = reinterpret_cast<uint32_t*>(p);
  uint32_t old = __atomic_load_n(u, __ATOMIC_RELAXED);
  for (;;) {
    const float nv = __uint_as_float(old) + v;
    uint32_t want = __float_as_uint(nv);
    if (__atomic_compare_exchange_n(u, &old, want, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return __uint_as_float(old);
  }
}
inline double atomicAdd(double* p, double v) {
  uint64_t* u = reinterpret_cast<uint64_t*>(p);
  uint64_t old = __atomic_load_n(u, __ATOMIC_RELAXED);
  for (;;) {
    const double nv = wavesim::from_bits<double>(old) + v;
    uint64_t want = wavesim::to_bits(nv);
    if (__atomic_compare_exchange_n(u, &old, want, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return wavesim::from_bits<double>(old);
  }
}
template <class T>
inline T atomicMin(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
  return old;
}
template <class T>
inline T atomicMax(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v > old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
  return old;
}
template <class T>
inline T atomicOr(T* p, T v) {
  return __atomic_fetch_or(p, v, __ATOMIC_RELAXED);
}
template <class T>
inline T atomicExch(T* p, T v) {
  return __atomic_exchange_n(p, v, __ATOMIC_RELAXED);
}

// integer min / max as the device headers overload them
inline uint32_t min(uint32_t a, uint32_t b) { return a < b ? a : b; }
inline uint32_t max(uint32_t a, uint32_t b) { return a > b ? a : b; }
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline uint64_t min(uint64_t a, uint64_t b) { return a < b ? a : b; }
inline uint64_t max(uint64_t a, uint64_t b) { return a > b ? a : b; }
inline unsigned long long min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
inline unsigned long long max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clz(int v) { return v ? __builtin_clz(unsigned(v)) : 32; }

// <cmath> classification functions as the device headers expose them
using std::isfinite;
using std::isinf;
using std::isnan;
using std::signbit;
