// normals_math.hpp -- per-point plane fit shared by the k-NN and the radius NormalEstimation kernels:
// covariance of a neighbourhood, closed-form smallest eigenvector, curvature, viewpoint flip.
// Float arithmetic in the reference's operation order (see the citations on each function).
#pragma once

#include <hip/hip_runtime.h>

namespace pclhip {

// computeMeanAndCovarianceMatrix, common/include/pcl/common/impl/centroid.hpp:581-650 (float,
// shifted by the first neighbour, neighbours in ascending distance order).
struct Cov {
  float a[9];
  float K[3];
  __device__ __forceinline__ void start(float x, float y, float z) {
#pragma unroll
    for (int i = 0; i < 9; ++i) a[i] = 0.0f;
    K[0] = x; K[1] = y; K[2] = z;
  }
  __device__ __forceinline__ void add(float px, float py, float pz) {
    const float x = __fsub_rn(px, K[0]), y = __fsub_rn(py, K[1]), z = __fsub_rn(pz, K[2]);
    a[0] = __fadd_rn(a[0], __fmul_rn(x, x));
    a[1] = __fadd_rn(a[1], __fmul_rn(x, y));
    a[2] = __fadd_rn(a[2], __fmul_rn(x, z));
    a[3] = __fadd_rn(a[3], __fmul_rn(y, y));
    a[4] = __fadd_rn(a[4], __fmul_rn(y, z));
    a[5] = __fadd_rn(a[5], __fmul_rn(z, z));
    a[6] = __fadd_rn(a[6], x);
    a[7] = __fadd_rn(a[7], y);
    a[8] = __fadd_rn(a[8], z);
  }
  // row-major symmetric 3x3
  __device__ __forceinline__ void finish(int count, float* c) {
    const float fc = float(count);
#pragma unroll
    for (int i = 0; i < 9; ++i) a[i] = __fdiv_rn(a[i], fc);
    c[0] = __fsub_rn(a[0], __fmul_rn(a[6], a[6]));
    c[1] = __fsub_rn(a[1], __fmul_rn(a[6], a[7]));
    c[2] = __fsub_rn(a[2], __fmul_rn(a[6], a[8]));
    c[4] = __fsub_rn(a[3], __fmul_rn(a[7], a[7]));
    c[5] = __fsub_rn(a[4], __fmul_rn(a[7], a[8]));
    c[8] = __fsub_rn(a[5], __fmul_rn(a[8], a[8]));
    c[3] = c[1]; c[6] = c[2]; c[7] = c[5];
  }
};

// common/include/pcl/common/impl/eigen.hpp:52-65
__device__ __forceinline__ void compute_roots2(float b, float c, float* r) {
  r[0] = 0.0f;
  float d = float(double(__fmul_rn(b, b)) - 4.0 * double(c));
  if (d < 0.0f) d = 0.0f;
  const float sd = sqrtf(d);
  r[2] = __fmul_rn(0.5f, __fadd_rn(b, sd));
  r[1] = __fmul_rn(0.5f, __fsub_rn(b, sd));
}

// common/include/pcl/common/impl/eigen.hpp:68-128
__device__ __forceinline__ void compute_roots(const float* m, float* roots) {
#define M_(i, j) m[(i)*3 + (j)]
  const float c0 = M_(0, 0) * M_(1, 1) * M_(2, 2) + 2.0f * M_(0, 1) * M_(0, 2) * M_(1, 2) -
                   M_(0, 0) * M_(1, 2) * M_(1, 2) - M_(1, 1) * M_(0, 2) * M_(0, 2) -
                   M_(2, 2) * M_(0, 1) * M_(0, 1);
  const float c1 = M_(0, 0) * M_(1, 1) - M_(0, 1) * M_(0, 1) + M_(0, 0) * M_(2, 2) - M_(0, 2) * M_(0, 2) +
                   M_(1, 1) * M_(2, 2) - M_(1, 2) * M_(1, 2);
  const float c2 = M_(0, 0) + M_(1, 1) + M_(2, 2);
#undef M_
  if (fabsf(c0) < FLT_EPSILON) {
    compute_roots2(c2, c1, roots);
  } else {
    const float s_inv3 = float(1.0 / 3.0);
    const float s_sqrt3 = sqrtf(3.0f);
    const float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.0f) a_over_3 = 0.0f;
    const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.0f) q = 0.0f;
    const float rho = sqrtf(-a_over_3);
    const float theta = atan2f(sqrtf(-q), half_b) * s_inv3;
    const float cos_theta = cosf(theta);
    const float sin_theta = sinf(theta);
    roots[0] = c2_over_3 + 2.0f * rho * cos_theta;
    roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    float t;
    if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
    if (roots[1] >= roots[2]) {
      t = roots[1]; roots[1] = roots[2]; roots[2] = t;
      if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
    }
    if (roots[0] <= 0) compute_roots2(c2, c1, roots);
  }
}

__device__ __forceinline__ void cross3(const float* a, const float* b, float* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

// common/include/pcl/common/impl/eigen.hpp:272-290
__device__ __forceinline__ void largest_eigvec(const float* sm, float* v) {
  float cp[3][3];
  cross3(sm + 0, sm + 3, cp[0]);
  cross3(sm + 0, sm + 6, cp[1]);
  cross3(sm + 3, sm + 6, cp[2]);
  float len = -1.0f;
  float bx = 0, by = 0, bz = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float l = sqrtf((cp[i][0] * cp[i][0] + cp[i][1] * cp[i][1]) + cp[i][2] * cp[i][2]);
    if (l > len) {
      len = l;
      bx = cp[i][0]; by = cp[i][1]; bz = cp[i][2];
    }
  }
  v[0] = bx / len; v[1] = by / len; v[2] = bz / len;
}

__device__ __forceinline__ void unit_orthogonal(const float* s, float* o) {
  const float prec = 1e-5f;
  const bool x_small = fabsf(s[0]) <= prec * fabsf(s[2]);
  const bool y_small = fabsf(s[1]) <= prec * fabsf(s[2]);
  if (!x_small || !y_small) {
    const float inv = 1.0f / sqrtf(s[0] * s[0] + s[1] * s[1]);
    o[0] = -s[1] * inv; o[1] = s[0] * inv; o[2] = 0.0f;
  } else {
    const float inv = 1.0f / sqrtf(s[1] * s[1] + s[2] * s[2]);
    o[0] = 0.0f; o[1] = -s[2] * inv; o[2] = s[1] * inv;
  }
}

// pcl::eigen33 (common/include/pcl/common/impl/eigen.hpp:295-325) + solvePlaneParameters
// (features/include/pcl/features/impl/feature.hpp:64-89)
__device__ __forceinline__ void solve_plane(const float* cov, float& nx, float& ny, float& nz, float& curv) {
  float scale = 0.0f;
#pragma unroll
  for (int i = 0; i < 9; ++i) scale = fmaxf(scale, fabsf(cov[i]));
  if (scale <= FLT_MIN) scale = 1.0f;
  float sm[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) sm[i] = cov[i] / scale;
  float ev[3], v[3];
  compute_roots(sm, ev);
  const float eigenvalue = ev[0] * scale;
  if ((ev[1] - ev[0]) > FLT_EPSILON) {
    sm[0] -= ev[0]; sm[4] -= ev[0]; sm[8] -= ev[0];
    largest_eigvec(sm, v);
  } else if ((ev[2] - ev[0]) > FLT_EPSILON) {
    sm[0] -= ev[2]; sm[4] -= ev[2]; sm[8] -= ev[2];
    float tmp[3];
    largest_eigvec(sm, tmp);
    unit_orthogonal(tmp, v);
  } else {
    v[0] = 1.0f; v[1] = 0.0f; v[2] = 0.0f;
  }
  nx = v[0]; ny = v[1]; nz = v[2];
  const float eig_sum = cov[0] + cov[4] + cov[8];
  curv = (eig_sum != 0) ? fabsf(eigenvalue / eig_sum) : 0.0f;
}

// features/include/pcl/features/normal_3d.h:169-188
__device__ __forceinline__ void flip_to_viewpoint(float px, float py, float pz, float vx, float vy, float vz,
                                                  float& nx, float& ny, float& nz) {
  vx -= px; vy -= py; vz -= pz;
  const float cos_theta = (vx * nx + vy * ny + vz * nz);
  if (cos_theta < 0) { nx *= -1; ny *= -1; nz *= -1; }
}

}  // namespace pclhip
