// pclhip_wave_reduce.hpp -- wavefront and row reductions of the traversal (traverse.hpp includes this by name through the
// include path).  THIS file: TEST INFRASTRUCTURE -- the host form for the emulation (tests/wavesim): the same reductions as shuffles;
// the Makefile here puts this directory in front of pcl_amd/csrc on the include path.
#pragma once

namespace pclhip {

__device__ __forceinline__ float wave_min_f(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ void wave_min3_max4(float& a0, float& a1, float& a2, float& b0, float& b1, float& b2,
                                               float& b3) {
  a0 = wave_min_f(a0); a1 = wave_min_f(a1); a2 = wave_min_f(a2);
  b0 = wave_max_f(b0); b1 = wave_max_f(b1); b2 = wave_max_f(b2); b3 = wave_max_f(b3);
}
__device__ __forceinline__ void row_max3_f(float& a, float& b, float& c) {
  for (int o = 8; o > 0; o >>= 1) {
    a = fmaxf(a, __shfl_xor(a, o)); b = fmaxf(b, __shfl_xor(b, o)); c = fmaxf(c, __shfl_xor(c, o));
  }
}
__device__ __forceinline__ void row_min3_f(float& a, float& b, float& c) {
  for (int o = 8; o > 0; o >>= 1) {
    a = fminf(a, __shfl_xor(a, o)); b = fminf(b, __shfl_xor(b, o)); c = fminf(c, __shfl_xor(c, o));
  }
}
#define PCLHIP_WAIT_VMCNT0() (void)0  // global_load_lds is immediate there

}  // namespace pclhip
