// Dev microbenchmark: how fast can a CU gather the 4 bilinear corner rows of a deformable-conv sample, and does the
// lane -> byte mapping matter?  Geometry of the 64-channel 112x200 DCN layers (16 frames, NHWC fp32), 128 pixels per
// workgroup, positions = pixel + tap + N(0, sigma) px.  Nothing but the loads (summed so they stay alive).
//   A: lane = (pixel, 8-channel unit): two 16-B loads at a 32-B lane stride per corner, 32-channel chunks  (the kernel today)
//   B: lane = (pixel, 16-B piece of a 128-B line): 8 lanes read one whole line per corner, 32-channel chunks
//   D: lane = (pixel, 16-B piece of the 256-B row): 16 lanes read both lines of a corner, whole tap at once
//   L: the tile's window staged in LDS once (32-channel halves), corners gathered with ds_read_b128 in mapping A
// hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip && ./gather_rate [sigma]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int B = 16, H = 112, W = 200, C = 64, M = B * H * W;

__device__ __forceinline__ int corner(const int* __restrict__ pos, int m, int tap) { return pos[m * 9 + tap]; }

__global__ __launch_bounds__(256, 2) void gather_a(const float* __restrict__ x, const int* __restrict__ pos, float* __restrict__ out) {
  const int tid = threadIdx.x, m0 = blockIdx.x * 128;
  f32x4 s = {0, 0, 0, 0};
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = m0 + (tid >> 2) + 64 * i;
        const float* a0 = x + corner(pos, m, tap) + ch * 32 + (tid & 3) * 8;
        const float* a[4] = {a0, a0 + C, a0 + W * C, a0 + W * C + C};
#pragma unroll
        for (int k = 0; k < 4; ++k) s += *reinterpret_cast<const f32x4*>(a[k]) + *reinterpret_cast<const f32x4*>(a[k] + 4);
      }
  out[blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(256, 2) void gather_b(const float* __restrict__ x, const int* __restrict__ pos, float* __restrict__ out) {
  const int tid = threadIdx.x, m0 = blockIdx.x * 128;
  f32x4 s = {0, 0, 0, 0};
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + (tid >> 3) + 32 * i;
        const float* a0 = x + corner(pos, m, tap) + ch * 32 + (tid & 7) * 4;
        s += *reinterpret_cast<const f32x4*>(a0) + *reinterpret_cast<const f32x4*>(a0 + C) +
             *reinterpret_cast<const f32x4*>(a0 + W * C) + *reinterpret_cast<const f32x4*>(a0 + W * C + C);
      }
  out[blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(256, 2) void gather_d(const float* __restrict__ x, const int* __restrict__ pos, float* __restrict__ out) {
  const int tid = threadIdx.x, m0 = blockIdx.x * 128;
  f32x4 s = {0, 0, 0, 0};
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + (tid >> 4) + 16 * i;
      const float* a0 = x + corner(pos, m, tap) + (tid & 15) * 4;
      s += *reinterpret_cast<const f32x4*>(a0) + *reinterpret_cast<const f32x4*>(a0 + C) +
           *reinterpret_cast<const f32x4*>(a0 + W * C) + *reinterpret_cast<const f32x4*>(a0 + W * C + C);
    }
  out[blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
}

// L: tile = 8 x 16 pixels of one image; window = (8 + 2R) x (16 + 2R) pixel rows of 32 channels (128 B + 16 B pad)
constexpr int R = 4, WH = 8 + 2 * R + 1, WW = 16 + 2 * R + 1, LROW = 128 + 16;
__global__ __launch_bounds__(256, 2) void gather_l(const float* __restrict__ x, const signed char* __restrict__ dyx, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char win[];
  const int tid = threadIdx.x;
  const int tiles_x = (W + 15) / 16, tiles_y = H / 8;
  const int b = blockIdx.x / (tiles_x * tiles_y), r = blockIdx.x % (tiles_x * tiles_y);
  const int y0 = (r / tiles_x) * 8, x0 = (r % tiles_x) * 16;
  f32x4 s = {0, 0, 0, 0};
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
    for (int i = tid; i < WH * WW * 8; i += 256) {
      const int row = i >> 3, u = i & 7;
      const int y = min(max(y0 - R + row / WW, 0), H - 1), xx = min(max(x0 - R + row % WW, 0), W - 1);
      *reinterpret_cast<f32x4*>(win + row * LROW + u * 16) =
          *reinterpret_cast<const f32x4*>(x + ((size_t)(b * H + y) * W + xx) * C + half * 32 + u * 4);
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int px = (tid >> 2) + 64 * i;
        const int py = y0 + (px >> 4), pxx = x0 + (px & 15);
        const int m = (b * H + min(py, H - 1)) * W + min(pxx, W - 1);
        const int d = dyx[(m * 9 + tap) * 2], e = dyx[(m * 9 + tap) * 2 + 1];
        const int wy = min(max((px >> 4) + R + tap / 3 - 1 + d, 0), WH - 2), wx = min(max((px & 15) + R + tap % 3 - 1 + e, 0), WW - 2);
        const unsigned char* a0 = win + (wy * WW + wx) * LROW + (tid & 3) * 32;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned char* a = a0 + ((k >> 1) * WW + (k & 1)) * LROW;
          s += *reinterpret_cast<const f32x4*>(a) + *reinterpret_cast<const f32x4*>(a + 16);
        }
      }
  }
  out[blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
}

// M: the layout planned for the window kernel.  Tile 16 x 16 pixels, window 25 x 25 rows of ONE 16-channel slice kept as
// 4 planes of 4 channels ([plane][row][16 B]: the 16 lanes of one tile row read 256 consecutive bytes), a lane = (pixel, half
// of the slice) reads 2 x 16 B per corner for 2 pixels - exactly the B fragments of v_mfma_f32_32x32x16_f16.
constexpr int MR = 4, MW = 16 + 2 * MR + 1, MROWS = MW * MW;
template <bool STAGE, bool GATHER>
__global__ __launch_bounds__(256, 2) void gather_m(const float* __restrict__ x, const signed char* __restrict__ dyx, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) unsigned char win[4 * MROWS * 16];
  __shared__ int desc[256 * 9];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int tiles_x = (W + 15) / 16, tiles_y = H / 16;
  const int b = blockIdx.x / (tiles_x * tiles_y), r = blockIdx.x % (tiles_x * tiles_y);
  const int y0 = (r / tiles_x) * 16, x0 = (r % tiles_x) * 16;
  for (int i = tid; i < 256 * 9; i += 256) {
    const int px = i / 9, tap = i % 9;
    const int py = y0 + (px >> 4), pxx = x0 + (px & 15);
    const int m = (b * H + min(py, H - 1)) * W + min(pxx, W - 1);
    const int d = dyx[(m * 9 + tap) * 2], e = dyx[(m * 9 + tap) * 2 + 1];
    const int wy = min(max((px >> 4) + MR + tap / 3 - 1 + d, 0), MW - 2), wx = min(max((px & 15) + MR + tap % 3 - 1 + e, 0), MW - 2);
    desc[i] = (wy * MW + wx) * 16;
  }
  f32x4 s = {0, 0, 0, 0};
  for (int slice = 0; slice < 4; ++slice) {
    __syncthreads();
    for (int i = tid; i < MROWS * 4; i += 256) {
      const int row = i >> 2, q = i & 3;
      const int y = y0 - MR + row / MW, xx = x0 - MR + row % MW;
      f32x4 v = {0, 0, 0, 0};
      if (STAGE && (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W)
        v = *reinterpret_cast<const f32x4*>(x + ((size_t)(b * H + y) * W + xx) * C + slice * 16 + q * 4);
      *reinterpret_cast<f32x4*>(win + (q * MROWS + row) * 16) = v;
    }
    __syncthreads();
    if (GATHER)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int px = (wave * 4 + ct * 2 + (li >> 4)) * 16 + (li & 15);
        const unsigned char* a0 = win + desc[px * 9 + tap] + (2 * h) * MROWS * 16;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned char* a = a0 + ((k >> 1) * MW + (k & 1)) * 16;
          s += *reinterpret_cast<const f32x4*>(a) + *reinterpret_cast<const f32x4*>(a + MROWS * 16);
        }
      }
  }
  out[blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
}

int main(int argc, char** argv) {
  const float sigma = argc > 1 ? atof(argv[1]) : 1.0f;
  std::vector<int> pos((size_t)M * 9);
  std::vector<signed char> dyx((size_t)M * 18);
  srand(1);
  auto gauss = [&]() { float u = (rand() + 1.0f) / (RAND_MAX + 2.0f), v = rand() / (float)RAND_MAX; return sqrtf(-2 * logf(u)) * cosf(6.2831853f * v); };
  for (int m = 0; m < M; ++m) {
    const int b = m / (H * W), y = (m / W) % H, xx = m % W;
    for (int t = 0; t < 9; ++t) {
      const int d = (int)floorf(gauss() * sigma), e = (int)floorf(gauss() * sigma);
      dyx[(m * 9 + t) * 2] = (signed char)d;
      dyx[(m * 9 + t) * 2 + 1] = (signed char)e;
      const int yy = std::min(std::max(y + t / 3 - 1 + d, 0), H - 2), xc = std::min(std::max(xx + t % 3 - 1 + e, 0), W - 2);
      pos[(size_t)m * 9 + t] = ((b * H + yy) * W + xc) * C;
    }
  }
  float *x, *out; int* dpos; signed char* ddyx;
  hipMalloc(&x, (size_t)M * C * 4); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&dpos, pos.size() * 4); hipMalloc(&ddyx, dyx.size());
  hipMemset(x, 0, (size_t)M * C * 4);
  hipMemcpy(dpos, pos.data(), pos.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(ddyx, dyx.data(), dyx.size(), hipMemcpyHostToDevice);
  const int lds = WH * WW * LROW;
  hipFuncSetAttribute((const void*)gather_l, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double gb = (double)M * 9 * 4 * C * 4 / 1e9;
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s  %7.1f us / launch   %6.2f TB/s of corner bytes   (%s)\n", name, ms * 50, gb / (ms / 20 * 1e-3) / 1e3, hipGetErrorString(hipGetLastError()));
  };
  printf("sigma %.2f px, %d pixels x 9 taps x 4 corners x 256 B = %.2f GB per launch; window LDS %d B\n", sigma, M, gb, lds);
  run("A  4 lanes x 2 x 16 B @32 B stride", [&] { hipLaunchKernelGGL(gather_a, dim3(M / 128), dim3(256), 0, 0, x, dpos, out); });
  run("B  8 lanes x 16 B = one line     ", [&] { hipLaunchKernelGGL(gather_b, dim3(M / 128), dim3(256), 0, 0, x, dpos, out); });
  run("D  16 lanes x 16 B = whole row   ", [&] { hipLaunchKernelGGL(gather_d, dim3(M / 128), dim3(256), 0, 0, x, dpos, out); });
  run("L  LDS window, ds_read_b128      ", [&] { hipLaunchKernelGGL(gather_l, dim3(B * (H / 8) * ((W + 15) / 16)), dim3(256), lds, 0, x, ddyx, out); });
  run("M  staging only                     ", [&] { hipLaunchKernelGGL((gather_m<true, false>), dim3(B * (H / 16) * ((W + 15) / 16)), dim3(256), 0, 0, x, ddyx, out); });
  run("M  gather only (window = zeros)     ", [&] { hipLaunchKernelGGL((gather_m<false, true>), dim3(B * (H / 16) * ((W + 15) / 16)), dim3(256), 0, 0, x, ddyx, out); });
  run("M  16x16 tile, 4-ch planes, lane=px ", [&] { hipLaunchKernelGGL((gather_m<true, true>), dim3(B * (H / 16) * ((W + 15) / 16)), dim3(256), 0, 0, x, ddyx, out); });
  return 0;
}
