// icp_xform.hpp -- how a search launch moves the working cloud (shared by search.hip and lane.hip).
#pragma once

#include "pclhip_internal.hpp"

namespace pclhip {

struct Mat34 {
  float m[12];  // rows 0..2 of the 4x4
};

// order 0: Eigen Matrix4f * Vector4f (registration/include/pcl/registration/impl/icp.hpp:49-111)
// order 1: Transformer<float>::se3 (common/include/pcl/common/impl/transforms.hpp:117-123)
__device__ __forceinline__ float xform_row(float r0, float r1, float r2, float r3, float x, float y, float z,
                                           int order) {
  if (order == 0)
    return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(r0, x), __fmul_rn(r1, y)), __fmul_rn(r2, z)), __fmul_rn(r3, 1.0f));
  return __fadd_rn(__fmul_rn(r0, x), __fadd_rn(__fmul_rn(r1, y), __fadd_rn(__fmul_rn(r2, z), r3)));
}

__device__ __forceinline__ bool in_region(const RegionBox& r, float x, float y, float z) {
  return x >= r.lo[0] && x < r.hi[0] && y >= r.lo[1] && y < r.hi[1] && z >= r.lo[2] && z < r.hi[2];
}

}  // namespace pclhip
