// Dev microbenchmark: issue cycles of v_mfma_f32_16x16x32_f16 and v_mfma_scale_f32_16x16x128_f8f6f4 (FP6 / FP8 operands) with
// register-resident operands, 32 independent accumulators per wave - the rates the heads' mx first layer is priced against.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_mx_rate mfma_mx_rate.hip && ./mfma_mx_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const int* __restrict__ src, float* __restrict__ out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  i32x8 a[4], b[8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) a[i][j] = src[(i * 8 + j) * 64 + lane];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) b[i][j] = src[((i + 4) * 8 + j) * 64 + lane];
  int sa = (src[lane] & 0x03030303) + 0x7c7c7c7c, sb = (src[64 + lane] & 0x03030303) + 0x7c7c7c7c;
  f32x4 acc[4][8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (MODE == 0) {
          const f16x8 fa = __builtin_bit_cast(f16x8, __builtin_shufflevector(a[i], a[i], 0, 1, 2, 3));
          const f16x8 fb = __builtin_bit_cast(f16x8, __builtin_shufflevector(b[j], b[j], 0, 1, 2, 3));
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[i][j], 0, 0, 0);
        } else if (MODE == 1) {
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i][j], 2, 2, 0, sa, 0, sb);   // FP6 x FP6
        } else if (MODE == 2) {
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i][j], 0, 0, 0, sa, 0, sb);   // FP8 x FP8
        } else if (MODE == 3) {
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i][j], 4, 4, 0, sa, 0, sb);   // FP4 x FP4
        } else {                                                                                                       // the head kernel's mix: 2 f16 : 1 fp6
          const f16x8 fa = __builtin_bit_cast(f16x8, __builtin_shufflevector(a[i], a[i], 0, 1, 2, 3));
          const f16x8 fb = __builtin_bit_cast(f16x8, __builtin_shufflevector(b[j], b[j], 0, 1, 2, 3));
          const f16x8 fc = __builtin_bit_cast(f16x8, __builtin_shufflevector(b[j], b[j], 4, 5, 6, 7));
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fc, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i][j], 2, 2, 0, sa, 0, sb);
        }
      }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  int* d; float* o; long long* c;
  hipMalloc(&d, 96 * 64 * 4 * 2); hipMalloc(&o, 4096 * 256 * 4); hipMalloc(&c, 8);
  int h[96 * 64 * 2];
  for (auto& v : h) v = rand() * 2654435761u;
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  const char* names[5] = {"16x16x32 f16", "16x16x128 scaled FP6 x FP6", "16x16x128 scaled FP8 x FP8", "16x16x128 scaled FP4 x FP4", "2 x f16 + 1 x scaled FP6 (per tile)"};
  const int iters = 2000;
  for (int blocks : {256, 512, 2048}) {          // 1 / 2 workgroups per CU resident (4 / 8 waves per CU), then a long grid
    for (int mode = 0; mode < 5; ++mode) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&] {
        switch (mode) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, o, c, iters); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, o, c, iters); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, o, c, iters); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, o, c, iters); break;
          default: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, o, c, iters); break;
        }
      };
      launch();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
      const double n_mfma = (double)iters * 32 * (mode == 4 ? 3 : 1);
      const double waves_per_simd = blocks >= 512 ? 2 : 1;
      printf("blocks %4d  %-38s %.3f ms  wave clock %.1f per MFMA (x %g waves/SIMD)  -> %.1f us per 1e6 MFMA-issues per SIMD\n", blocks, names[mode], ms,
             cy / n_mfma, waves_per_simd, ms * 1e3 / (n_mfma * (blocks / 256.0) / 1e6) );
    }
  }
  return 0;
}
