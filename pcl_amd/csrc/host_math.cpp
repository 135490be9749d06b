// host_math.cpp -- host instantiation of the closed forms in closed_forms.hpp (the device twin is
// icp_solve_kernel in icp_loop.hip).  Tiny, latency-bound, fp64.
#include "closed_forms.hpp"
#include "pclhip_internal.hpp"

namespace pclhip {

void solve_point_to_plane(const double* s, float* T) { cf::solve_point_to_plane(s, T); }
void solve_symmetric(const double* s, float* T) { cf::solve_symmetric(s, T); }
void solve_point_to_point(const double* s, float* T) { cf::solve_point_to_point(s, T); }
void mat4_mul_f32(const float* A, const float* B, float* C) { cf::mat4_mul_f32(A, B, C); }

}  // namespace pclhip
