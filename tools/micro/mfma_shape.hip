// Dev microbenchmark: dense f16 MFMA rate of the two gfx950 shapes on RANDOM data with every operand re-read
// from LDS (the regime of the head / conv kernels).  hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// wave tile 64 x 128 (as the head kernel): 32x32x16: 2 x 4 tiles, per k16: 2 A + 4 B fragments, 8 MFMAs
template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(const f16x8* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ f16x8 lds[64 * 48];                     // 48 KiB of fragments
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 64 * 48; i += 256) lds[i] = src[(blockIdx.x % 64) * 64 * 48 + i];
  __syncthreads();
  if (SHAPE == 32) {
    f32x16 acc[2][4];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        f16x8 A[2], B[4];
#pragma unroll
        for (int a = 0; a < 2; ++a) A[a] = lds[((it * 3 + ks * 6 + a) & 31) * 64 + lane];
#pragma unroll
        for (int b = 0; b < 4; ++b) B[b] = lds[((it * 3 + ks * 6 + 2 + b) & 31) * 64 + lane];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[a], B[b], acc[a][b], 0, 0, 0);
      }
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 256 + tid] = s;
  } else {
    // same wave tile from 16x16x32 tiles: 4 x 8 tiles, per k32: 4 A + 8 B fragments, 32 MFMAs (= 2 k16 steps of work)
    f32x4 acc[4][8];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 8; ++b) for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        f16x8 A[4], B[8];
#pragma unroll
        for (int a = 0; a < 4; ++a) A[a] = lds[((it * 3 + ks * 12 + a) & 31) * 64 + lane];
#pragma unroll
        for (int b = 0; b < 8; ++b) B[b] = lds[((it * 3 + ks * 12 + 4 + b) & 31) * 64 + lane];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 8; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[a], B[b], acc[a][b], 0, 0, 0);
      }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 8; ++b) for (int r = 0; r < 4; ++r) s += acc[a][b][r];
    out[blockIdx.x * 256 + tid] = s;
  }
}

int main() {
  const int blocks = 512 * 8, iters = 400;
  std::vector<_Float16> h(64 * 64 * 48 * 8);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.25f);
  f16x8* d; float* o;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, blocks * 256 * 4);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop = (double)blocks * 4 * iters * 8 * 8 * 32.0 * 32 * 16 * 2;     // per launch (both kernels do the same work)
  for (int rep = 0; rep < 3; ++rep)
    for (int shape : {32, 16}) {
      for (int w = 0; w < 3; ++w) {
        if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
        else hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
      }
      hipEventRecord(e0);
      const int n = 5;
      for (int w = 0; w < n; ++w) {
        if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
        else hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("shape %2d: %.3f ms per launch = %.0f TFLOP/s\n", shape, ms / n, flop / (ms / n * 1e-3) / 1e12);
    }
  return 0;
}
