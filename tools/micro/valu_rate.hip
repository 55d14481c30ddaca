// Dev microbenchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU instructions the f16x3 kernels
// lean on.  2 waves per SIMD (the occupancy of the MFMA kernels), 8 independent chains per wave.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float seed) {
  float a[8], b[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; b[i] = seed * 0.5f + i; u[i] = threadIdx.x * 77 + i; }
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[8], q[8];
  for (int i = 0; i < 8; ++i) { p[i] = f2{a[i], b[i]}; q[i] = f2{b[i], a[i]}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (OP == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(q[i]));
        REP8(X)
#undef X
      } else if (OP == 2) {
#define X(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(b[i]));
        REP8(X)
#undef X
      } else if (OP == 3) {
#define X(i) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(a[i]));
        REP8(X)
#undef X
      } else if (OP == 4) {
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
        REP8(X)
#undef X
      } else if (OP == 5) {
#define X(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(u[i]));
        REP8(X)
#undef X
      } else if (OP == 6) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
        REP8(X)
#undef X
      } else if (OP == 7) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
        REP8(X)
#undef X
      } else if (OP == 8) {
#define X(i) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[i]) : "v"(u[i]));
        REP8(X)
#undef X
      } else if (OP == 9) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
        REP8(X)
#undef X
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + b[i] + p[i][0] + p[i][1] + (float)u[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, float* out) {
  const int iters = 20000, blocks = 512;       // 2 workgroups of 4 waves per CU: 2 waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double inst_per_simd = 2.0 * iters * 32;     // 2 waves x instructions
  printf("%-22s %7.2f ns per wave-instruction per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
}
int main() {
  float* out; hipMalloc(&out, 512 * 256 * 4);
  run<0>("v_fma_f32", out); run<1>("v_pk_fma_f32", out); run<6>("v_pk_mul_f32", out); run<7>("v_pk_add_f32", out);
  run<2>("v_cvt_pk_f16_f32", out); run<3>("v_fma_mixlo_f16", out); run<4>("v_med3_f32", out); run<5>("v_cvt_f32_f16", out);
  run<8>("v_cvt_f32_f16_sdwa", out); run<9>("v_mov_b32", out);
  return 0;
}
