// closed_forms.hpp -- the tiny fp64 closed forms of the ICP loop, written once and compiled for BOTH
// sides: the host (pclhip_solve_transformation, the host-driven loop used with rejectors) and the device
// (icp_solve_kernel, which closes the iteration on the GPU so the loop never waits for the host).
//   * 6x6 normal-system solve + constructTransformationMatrix
//     (registration/include/pcl/registration/impl/transformation_estimation_point_to_plane_lls.hpp:132-163,247-268)
//   * the symmetric objective's solve
//     (impl/transformation_estimation_symmetric_point_to_plane_lls.hpp:128-147,192-196)
//   * umeyama from raw sums (common/include/pcl/common/impl/eigen.hpp:675-738)
//   * float 4x4 product for final_transformation_ (registration/include/pcl/registration/impl/icp.hpp:223)
//   * DefaultConvergenceCriteria::hasConverged (impl/default_convergence_criteria.hpp:49-140)
// Everything is plain IEEE double arithmetic in a fixed order (-ffp-contract=off on both sides); the only
// library calls are sqrt (correctly rounded on both sides) and sin/cos (device and host libm can differ in
// the last ulp of a double, far below the float the result is rounded to).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/pclhip.h"

#define PCLHIP_HD __host__ __device__ inline

namespace pclhip {
namespace cf {

template <class T>
PCLHIP_HD void swap_(T& a, T& b) {
  const T t = a;
  a = b;
  b = t;
}

// Doolittle LU with partial pivoting on a 6x6, then forward/back substitution.  Every array index is a compile-time
// constant after unrolling (the pivot row is swapped in by predicated exchanges with each candidate row), so on
// the device the system lives in registers: a row index known only at run time would put it in scratch memory,
// and the one thread that closes an ICP iteration would wait out a memory round trip per element.
PCLHIP_HD bool lu_solve6(double A[6][6], double b[6], double x[6]) {
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    int p = c;
    double best = fabs(A[c][c]);
#pragma unroll
    for (int r = c + 1; r < 6; ++r)
      if (fabs(A[r][c]) > best) {
        best = fabs(A[r][c]);
        p = r;
      }
    if (best == 0.0) return false;
#pragma unroll
    for (int r = c + 1; r < 6; ++r) {
      const bool sw = p == r;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const double u = A[c][j], v = A[r][j];
        A[c][j] = sw ? v : u;
        A[r][j] = sw ? u : v;
      }
      const double u = b[c], v = b[r];
      b[c] = sw ? v : u;
      b[r] = sw ? u : v;
    }
#pragma unroll
    for (int r = c + 1; r < 6; ++r) {
      const double f = A[r][c] / A[c][c];
      A[r][c] = f;
#pragma unroll
      for (int j = c + 1; j < 6; ++j) A[r][j] -= f * A[c][j];
      b[r] -= f * b[c];
    }
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double s = b[i];
#pragma unroll
    for (int j = i + 1; j < 6; ++j) s -= A[i][j] * x[j];
    x[i] = s / A[i][i];
  }
  return true;
}

// cyclic Jacobi eigen-decomposition of a symmetric 3x3: A = V diag(w) V^T
PCLHIP_HD void jacobi_eig3(double A[3][3], double V[3][3], double w[3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
  // Stop when the off-diagonal part is 1e-30 of the diagonal: fourteen orders below what a double can tell apart, reached
  // after ~5 sweeps (the convergence is quadratic).  Until round 5 the loop ran until the off-diagonals underflowed
  // (< 1e-300: ~9 sweeps) -- four sweeps of divisions and square roots that change no bit of the result, and the
  // device runs them on ONE lane at full instruction latency: icp_finalize_kernel 78 us per iteration of a point-to-point
  // registration (config 2), a third of its step.
  const double scale = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
  for (int sweep = 0; sweep < 64; ++sweep) {
    const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    if (off < 1e-300 || off <= 1e-30 * scale) break;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // A <- A * J
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // A <- J^T * A
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < 3; ++i) w[i] = A[i][i];
}

PCLHIP_HD double det3(const double M[3][3]) {
  return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
         M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}

PCLHIP_HD void cross3(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

// R = U S V^T of sigma via the eigen-decomposition of sigma^T sigma (V, singular values) and
// U = sigma V / s, with the umeyama reflection fix on the smallest singular direction.
PCLHIP_HD void rotation_from_sigma(const double sigma[3][3], double R[3][3]) {
  double AtA[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += sigma[k][i] * sigma[k][j];
      AtA[i][j] = s;
    }
  double V[3][3], w[3];
  jacobi_eig3(AtA, V, w);
  // descending eigenvalues: selection by exchanges of whole (value, vector) pairs -- constant indices only
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = i + 1; j < 3; ++j) {
      const bool sw = w[j] > w[i];
      const double wi = w[i], wj = w[j];
      w[i] = sw ? wj : wi;
      w[j] = sw ? wi : wj;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double vi = V[r][i], vj = V[r][j];
        V[r][i] = sw ? vj : vi;
        V[r][j] = sw ? vi : vj;
      }
    }
  double Vs[3][3], sv[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    sv[c] = sqrt(w[c] > 0 ? w[c] : 0.0);
#pragma unroll
    for (int r = 0; r < 3; ++r) Vs[r][c] = V[r][c];
  }
  if (det3(Vs) < 0)  // keep V a proper rotation; the sign moves into U, U S V^T is unchanged
    for (int r = 0; r < 3; ++r) Vs[r][2] = -Vs[r][2];
  double U[3][3];
  const double tol = 1e-13 * (sv[0] > 0 ? sv[0] : 1.0);
  for (int c = 0; c < 3; ++c) {
    double u[3];
    for (int r = 0; r < 3; ++r) u[r] = sigma[r][0] * Vs[0][c] + sigma[r][1] * Vs[1][c] + sigma[r][2] * Vs[2][c];
    const double n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    for (int r = 0; r < 3; ++r) U[r][c] = (sv[c] > tol && n > 0) ? u[r] / n : 0.0;
  }
  // complete rank-deficient cases with cross products (right-handed completion)
  if (!(sv[0] > tol)) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) U[i][j] = (i == j) ? 1.0 : 0.0;
  } else {
    double u0[3] = {U[0][0], U[1][0], U[2][0]}, u1[3], u2[3];
    if (!(sv[1] > tol)) {
      int mi = 0;
      double least = fabs(u0[0]);
#pragma unroll
      for (int d = 1; d < 3; ++d)
        if (fabs(u0[d]) < least) {
          least = fabs(u0[d]);
          mi = d;
        }
      const double e[3] = {mi == 0 ? 1.0 : 0.0, mi == 1 ? 1.0 : 0.0, mi == 2 ? 1.0 : 0.0};
      cross3(u0, e, u1);
      const double n = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
      for (int d = 0; d < 3; ++d) U[d][1] = u1[d] / n;
    }
    for (int d = 0; d < 3; ++d) u1[d] = U[d][1];
    if (!(sv[2] > tol)) {
      cross3(u0, u1, u2);
      for (int d = 0; d < 3; ++d) U[d][2] = u2[d];
    }
  }
  // eigen.hpp:716-724: S = diag(1,1,-1) if det(U) det(V) < 0
  const double sgn = (det3(U) * det3(Vs) < 0) ? -1.0 : 1.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i][j] = U[i][0] * Vs[j][0] + U[i][1] * Vs[j][1] + sgn * U[i][2] * Vs[j][2];
}

PCLHIP_HD void load_normal_system(const double* s, double A[6][6], double b[6]) {
  int k = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) {
      A[i][j] = s[k];
      A[j][i] = s[k];
      ++k;
    }
  for (int i = 0; i < 6; ++i) b[i] = s[21 + i];
}

PCLHIP_HD void zero16(float* T) {
  for (int i = 0; i < 16; ++i) T[i] = 0.0f;
}

PCLHIP_HD void solve_point_to_plane(const double* s, float* T) {
  double A[6][6], b[6], x[6] = {0, 0, 0, 0, 0, 0};
  load_normal_system(s, A, b);
  if (!lu_solve6(A, b, x))
    for (int i = 0; i < 6; ++i) x[i] = nan("");  // singular system: Eigen's inverse() yields non-finite too
  // each sine and cosine once (the products below are the reference's expressions, same operand order)
  const double sa = sin(x[0]), ca = cos(x[0]), sb = sin(x[1]), cb = cos(x[1]), sg = sin(x[2]), cg = cos(x[2]);
  zero16(T);
  T[0] = float(cg * cb);
  T[1] = float(-sg * ca + cg * sb * sa);
  T[2] = float(sg * sa + cg * sb * ca);
  T[4] = float(sg * cb);
  T[5] = float(cg * ca + sg * sb * sa);
  T[6] = float(-cg * sa + sg * sb * ca);
  T[8] = float(-sb);
  T[9] = float(cb * sa);
  T[10] = float(cb * ca);
  T[3] = float(x[3]);
  T[7] = float(x[4]);
  T[11] = float(x[5]);
  T[15] = 1.0f;
}

// T = Rz Ry Rx * Translation(t) * Rz Ry Rx = [R R | R t] with R = Rz(x2) Ry(x1) Rx(x0)
PCLHIP_HD void solve_symmetric(const double* s, float* T) {
  double A[6][6], b[6], x[6] = {0, 0, 0, 0, 0, 0};
  load_normal_system(s, A, b);
  if (!lu_solve6(A, b, x))
    for (int i = 0; i < 6; ++i) x[i] = nan("");
  const double ca = cos(x[0]), sa = sin(x[0]), cb = cos(x[1]), sb = sin(x[1]), cg = cos(x[2]), sg = sin(x[2]);
  const double R[3][3] = {{cg * cb, -sg * ca + cg * sb * sa, sg * sa + cg * sb * ca},
                          {sg * cb, cg * ca + sg * sb * sa, -cg * sa + sg * sb * ca},
                          {-sb, cb * sa, cb * ca}};
  zero16(T);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      double v = 0.0;
      for (int m = 0; m < 3; ++m) v += R[i][m] * R[m][j];
      T[4 * i + j] = float(v);
    }
    T[4 * i + 3] = float(R[i][0] * x[3] + R[i][1] * x[4] + R[i][2] * x[5]);
  }
  T[15] = 1.0f;
}

PCLHIP_HD void solve_point_to_point(const double* s, float* T) {
  const double n = s[28];
  double sm[3], dm[3], sigma[3][3], R[3][3];
  for (int d = 0; d < 3; ++d) {
    sm[d] = s[d] / n;
    dm[d] = s[3 + d] / n;
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) sigma[i][j] = s[6 + 3 * i + j] / n - dm[i] * sm[j];
  rotation_from_sigma(sigma, R);
  zero16(T);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[4 * i + j] = float(R[i][j]);
    T[4 * i + 3] = float(dm[i] - (R[i][0] * sm[0] + R[i][1] * sm[1] + R[i][2] * sm[2]));
  }
  T[15] = 1.0f;
}

PCLHIP_HD void solve(const double* sums, int mode, float* T) {
  if (mode == PCLHIP_ICP_POINT_TO_PLANE)
    solve_point_to_plane(sums, T);
  else if (mode == PCLHIP_ICP_SYMMETRIC)
    solve_symmetric(sums, T);
  else
    solve_point_to_point(sums, T);
}

// float product in Eigen's coefficient order ((a0*b0 + a1*b1) + a2*b2) + a3*b3; C may alias A or B.
// (The translation unit is compiled with -ffp-contract=off, so no product is fused into a sum.)
PCLHIP_HD void mat4_mul_f32(const float* A, const float* B, float* C) {
  float R[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      const float t0 = A[4 * i + 0] * B[0 * 4 + j];
      const float t1 = A[4 * i + 1] * B[1 * 4 + j];
      const float t2 = A[4 * i + 2] * B[2 * 4 + j];
      const float t3 = A[4 * i + 3] * B[3 * 4 + j];
      float r = t0 + t1;
      r = r + t2;
      r = r + t3;
      R[4 * i + j] = r;
    }
  for (int i = 0; i < 16; ++i) C[i] = R[i];
}

// ---- DefaultConvergenceCriteria ---------------------------------------------------------------------
enum ConvergenceState {
  NOT_CONVERGED = 0, ITERATIONS, TRANSFORM, ABS_MSE, REL_MSE, NO_CORRESPONDENCES, FAILURE_AFTER_MAX_ITERATIONS
};

struct Criteria {            // parameters (ICP sets them from its own members, impl/icp.hpp:157-161)
  int max_iterations;
  int failure_after_max_iterations;
  int max_iterations_similar_transforms;
  int min_number_correspondences;
  double rotation_threshold;     // cos(angle); 0.99999 unless transformation_rotation_epsilon is set
  double translation_threshold;  // transformation_epsilon
  double mse_threshold_relative; // euclidean_fitness_epsilon
  double mse_threshold_absolute;
};

struct CriteriaState {       // persists across align() calls on the same object, as in the reference
  double prev_mse;
  int iterations_similar_transforms;
  int convergence_state;
};

// hasConverged (impl/default_convergence_criteria.hpp:49-140); nr_iterations already counts this iteration.
// Returns true when the loop ends with converged_ = true; the caller leaves the loop whenever
// st.convergence_state != NOT_CONVERGED.
PCLHIP_HD bool has_converged(const Criteria& c, CriteriaState& st, int nr_iterations, const float* Tk, double mse) {
  if (st.convergence_state != NOT_CONVERGED) {
    st.iterations_similar_transforms = 0;
    st.convergence_state = NOT_CONVERGED;
  }
  bool is_similar = false;
  if (nr_iterations >= c.max_iterations) {
    if (!c.failure_after_max_iterations) {
      st.convergence_state = ITERATIONS;
      return true;
    }
    st.convergence_state = FAILURE_AFTER_MAX_ITERATIONS;
  }
  const double cos_angle = 0.5 * double(Tk[0] + Tk[5] + Tk[10] - 1);
  const double translation_sqr = double(Tk[3] * Tk[3] + Tk[7] * Tk[7] + Tk[11] * Tk[11]);
  if (cos_angle >= c.rotation_threshold && translation_sqr <= c.translation_threshold) {
    if (st.iterations_similar_transforms >= c.max_iterations_similar_transforms) {
      st.convergence_state = TRANSFORM;
      return true;
    }
    is_similar = true;
  }
  if (fabs(mse - st.prev_mse) < c.mse_threshold_absolute) {
    if (st.iterations_similar_transforms >= c.max_iterations_similar_transforms) {
      st.convergence_state = ABS_MSE;
      return true;
    }
    is_similar = true;
  }
  if (fabs(mse - st.prev_mse) / st.prev_mse < c.mse_threshold_relative) {
    if (st.iterations_similar_transforms >= c.max_iterations_similar_transforms) {
      st.convergence_state = REL_MSE;
      return true;
    }
    is_similar = true;
  }
  if (is_similar)
    ++st.iterations_similar_transforms;
  else
    st.iterations_similar_transforms = 0;
  st.prev_mse = mse;
  return false;
}

}  // namespace cf
}  // namespace pclhip
