// VALU issue-cost microbenchmark on gfx950: cycles per wave-instruction per SIMD for the op mixes
// the leaf loop can be built from.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 4096
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
  float x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x2}, p5 = {x3, x4}, p6 = {x5, x6}, p7 = {x7, x0};
  v2f pb = {b, b};
  unsigned long long u0 = threadIdx.x, u1 = u0 + 1;
  for (int i = 0; i < ITERS; ++i) {
    if (OP == 0) {  // v_add_f32
      x0 = __fadd_rn(x0, b); x1 = __fadd_rn(x1, b); x2 = __fadd_rn(x2, b); x3 = __fadd_rn(x3, b);
      x4 = __fadd_rn(x4, b); x5 = __fadd_rn(x5, b); x6 = __fadd_rn(x6, b); x7 = __fadd_rn(x7, b);
    } else if (OP == 1) {  // v_mul_f32
      x0 = __fmul_rn(x0, b); x1 = __fmul_rn(x1, b); x2 = __fmul_rn(x2, b); x3 = __fmul_rn(x3, b);
      x4 = __fmul_rn(x4, b); x5 = __fmul_rn(x5, b); x6 = __fmul_rn(x6, b); x7 = __fmul_rn(x7, b);
    } else if (OP == 2) {  // v_fma_f32
      x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);
      x4 = fmaf(x4, b, a); x5 = fmaf(x5, b, a); x6 = fmaf(x6, b, a); x7 = fmaf(x7, b, a);
    } else if (OP == 3) {  // v_pk_add_f32
      p0 = p0 + pb; p1 = p1 + pb; p2 = p2 + pb; p3 = p3 + pb; p4 = p4 + pb; p5 = p5 + pb; p6 = p6 + pb; p7 = p7 + pb;
    } else if (OP == 4) {  // v_pk_mul_f32
      p0 = p0 * pb; p1 = p1 * pb; p2 = p2 * pb; p3 = p3 * pb; p4 = p4 * pb; p5 = p5 * pb; p6 = p6 * pb; p7 = p7 * pb;
    } else if (OP == 5) {  // v_min_f32
      x0 = fminf(x0, b); x1 = fminf(x1, a); x2 = fminf(x2, b); x3 = fminf(x3, a);
      x4 = fminf(x4, b); x5 = fminf(x5, a); x6 = fminf(x6, b); x7 = fminf(x7, a);
      asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else if (OP == 6) {  // v_cmp_lt_u64 + 2 cndmask
      bool t = u0 < u1; u1 = t ? u0 + i : u1; u0 += 3;
      bool t2 = u1 < u0; u0 = t2 ? u1 + i : u0; u1 += 5;
    } else if (OP == 7) {  // v_max3_f32
      x0 = __builtin_fmaxf(__builtin_fmaxf(x0, x1), b); x2 = __builtin_fmaxf(__builtin_fmaxf(x2, x3), b);
      x4 = __builtin_fmaxf(__builtin_fmaxf(x4, x5), b); x6 = __builtin_fmaxf(__builtin_fmaxf(x6, x7), b);
      asm volatile("" : "+v"(x0), "+v"(x2), "+v"(x4), "+v"(x6));
    }
    if (OP == 3 || OP == 4) asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));
    else if (OP < 3) asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + float(u0 + u1);
}
template <int OP> void run(const char* name, int ops_per_iter, int blocks_per_cu) {
  float* d; hipMalloc(&d, 256 * 256 * 16 * sizeof(float));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  int grid = 256 * blocks_per_cu;
  k<OP><<<grid, 256>>>(d, 1.0f, 1.0001f);
  hipEventRecord(a); k<OP><<<grid, 256>>>(d, 1.0f, 1.0001f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double wave_instr_per_simd = double(ITERS) * ops_per_iter * (grid * 4) / 1024.0;
  printf("%-14s blocks/CU %d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, blocks_per_cu, ms,
         ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4);
  hipFree(d);
}
int main() {
  for (int bpc : {1, 2, 4}) {
    run<0>("v_add_f32", 8, bpc); run<1>("v_mul_f32", 8, bpc); run<2>("v_fma_f32", 8, bpc);
    run<3>("v_pk_add_f32", 8, bpc); run<4>("v_pk_mul_f32", 8, bpc); run<5>("v_min_f32", 8, bpc);
    run<6>("cmp_u64+cnd", 10, bpc); run<7>("v_max3_f32", 4, bpc);
  }
  return 0;
}
