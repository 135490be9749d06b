// pclhip_wave_reduce.hpp -- wavefront and row reductions of the traversal (traverse.hpp includes this by name through the
// include path).  THIS file: the product's form, DPP butterflies in inline assembly for gfx950 (see traverse.hpp for why).
#pragma once

namespace pclhip {

#define PCLHIP_DPP_STEPS(OP)                                                                 \
  "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"          \
  "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"          \
  "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"              \
  "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"                   \
  "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                 \
  "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                 \
  "s_nop 1"
__device__ __forceinline__ float wave_min_f(float v) {
  asm volatile(PCLHIP_DPP_STEPS("v_min_f32_dpp") : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_f(float v) {
  asm volatile(PCLHIP_DPP_STEPS("v_max_f32_dpp") : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#undef PCLHIP_DPP_STEPS
// three minima and four maxima at once (the query group's box and the wave radius): the seven chains
// interleave, so no wait states are needed
#define PCLHIP_DPP_STEP7(CTRL)                              \
  "v_min_f32_dpp %0, %0, %0 " CTRL "\n\t"                   \
  "v_min_f32_dpp %1, %1, %1 " CTRL "\n\t"                   \
  "v_min_f32_dpp %2, %2, %2 " CTRL "\n\t"                   \
  "v_max_f32_dpp %3, %3, %3 " CTRL "\n\t"                   \
  "v_max_f32_dpp %4, %4, %4 " CTRL "\n\t"                   \
  "v_max_f32_dpp %5, %5, %5 " CTRL "\n\t"                   \
  "v_max_f32_dpp %6, %6, %6 " CTRL "\n\t"
__device__ __forceinline__ void wave_min3_max4(float& a0, float& a1, float& a2, float& b0, float& b1, float& b2,
                                               float& b3) {
  asm volatile("s_nop 1\n\t" PCLHIP_DPP_STEP7("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                   PCLHIP_DPP_STEP7("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                       PCLHIP_DPP_STEP7("row_half_mirror row_mask:0xf bank_mask:0xf")
                           PCLHIP_DPP_STEP7("row_mirror row_mask:0xf bank_mask:0xf")
                               PCLHIP_DPP_STEP7("row_bcast:15 row_mask:0xa bank_mask:0xf")
                                   PCLHIP_DPP_STEP7("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
  a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a0), 63));
  a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a1), 63));
  a2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a2), 63));
  b0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b0), 63));
  b1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b1), 63));
  b2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b2), 63));
  b3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b3), 63));
}
#undef PCLHIP_DPP_STEP7
// reductions over every ROW of 16 lanes (each lane ends up with its row's result): three chains at once, which interleave,
// so no wait states are needed between the steps
#define PCLHIP_ROW_STEP3(O0, O1, O2, CTRL) \
  O0 " %0, %0, %0 " CTRL "\n\t"           \
  O1 " %1, %1, %1 " CTRL "\n\t"           \
  O2 " %2, %2, %2 " CTRL "\n\t"
#define PCLHIP_ROW_REDUCE3(O0, O1, O2)                                                       \
  "s_nop 1\n\t" PCLHIP_ROW_STEP3(O0, O1, O2, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
      PCLHIP_ROW_STEP3(O0, O1, O2, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")           \
          PCLHIP_ROW_STEP3(O0, O1, O2, "row_half_mirror row_mask:0xf bank_mask:0xf")           \
              PCLHIP_ROW_STEP3(O0, O1, O2, "row_mirror row_mask:0xf bank_mask:0xf") "s_nop 1"
__device__ __forceinline__ void row_max3_f(float& a, float& b, float& c) {
  asm volatile(PCLHIP_ROW_REDUCE3("v_max_f32_dpp", "v_max_f32_dpp", "v_max_f32_dpp") : "+v"(a), "+v"(b), "+v"(c));
}
__device__ __forceinline__ void row_min3_f(float& a, float& b, float& c) {
  asm volatile(PCLHIP_ROW_REDUCE3("v_min_f32_dpp", "v_min_f32_dpp", "v_min_f32_dpp") : "+v"(a), "+v"(b), "+v"(c));
}
#undef PCLHIP_ROW_REDUCE3
#undef PCLHIP_ROW_STEP3
#define PCLHIP_WAIT_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

}  // namespace pclhip
