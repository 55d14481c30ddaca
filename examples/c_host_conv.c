/* A host that is NOT Python driving libcfhip.so through its C ABI alone: one BasicBlock convolution of the reference
 * (3x3, 64 -> 64, BatchNorm folded, ReLU: model/networks/dla.py:42-62) packed by cf_pack_conv_f16x3, run by cf_conv3x3_f16x3
 * on the GPU and checked against a plain double-precision convolution on the CPU.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/c_host_conv.c -o /tmp/c_host_conv \
 *       -Lcenterfusiondetect3d_amd -lcfhip -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/centerfusiondetect3d_amd -Wl,-rpath,/opt/rocm/lib
 *   /tmp/c_host_conv            -> "max |err| / max |ref| = ...", exit status 0 when below 2e-6
 * (tests/test_gpu_c_host.py builds and runs it.) */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "cf_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_CF(x) do { int rc_ = (x); if (rc_ != CF_OK) { fprintf(stderr, "%s: %d %s\n", #x, rc_, cf_last_error()); return 3; } } while (0)

static unsigned long long rng = 0x9E3779B97F4A7C15ull;
static float frand(void) {                          /* uniform in [-1, 1) */
  rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
  return (float)((double)(rng >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}

int main(void) {
  enum { B = 2, H = 19, W = 37, C = 64, N = 64 };
  const long M = (long)B * H * W;
  float* w = malloc(sizeof(float) * N * C * 9);
  float *gamma = malloc(4 * N), *beta = malloc(4 * N), *mean = malloc(4 * N), *var = malloc(4 * N);
  float* x = malloc(sizeof(float) * M * C);
  for (long i = 0; i < (long)N * C * 9; ++i) w[i] = frand() * 0.06f;
  for (int n = 0; n < N; ++n) { gamma[n] = 1.0f + 0.3f * frand(); beta[n] = 0.2f * frand(); mean[n] = 0.3f * frand(); var[n] = 0.8f + 0.5f * frand(); }
  for (long i = 0; i < M * C; ++i) { const float v = frand() * 2.0f; x[i] = v > 0.0f ? v : 0.0f; }   /* a post-ReLU map, NHWC */

  /* ---- pack (host) */
  cf_pack_src src = {C, C, 0};
  cf_pack_conv_desc d;
  memset(&d, 0, sizeof d);
  d.weight = w;
  d.bn.gamma = gamma; d.bn.beta = beta; d.bn.mean = mean; d.bn.var = var; d.bn.eps = 1e-5f;
  d.cout = N; d.kh = 3; d.kw = 3; d.stride = 1; d.pad = -1; d.dilation = 1;
  d.src = &src; d.n_src = 1;
  cf_pack_info info;
  CHECK_CF(cf_pack_conv_f16x3_info(&d, &info));
  void* wpk = malloc(info.weight_bytes);
  cf_slot* slots = malloc(sizeof(cf_slot) * info.n_slots);
  float* bias = malloc(sizeof(float) * info.n_pad);
  CHECK_CF(cf_pack_conv_f16x3(&d, wpk, slots, bias, &info));

  /* ---- device buffers */
  void *dw, *ds, *db, *dx, *dout;
  CHECK_HIP(hipMalloc(&dw, info.weight_bytes));
  CHECK_HIP(hipMalloc(&ds, sizeof(cf_slot) * info.n_slots));
  CHECK_HIP(hipMalloc(&db, sizeof(float) * info.n_pad));
  CHECK_HIP(hipMalloc(&dx, sizeof(float) * M * C));
  CHECK_HIP(hipMalloc(&dout, sizeof(float) * M * N));
  CHECK_HIP(hipMemcpy(dw, wpk, info.weight_bytes, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(ds, slots, sizeof(cf_slot) * info.n_slots, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(db, bias, sizeof(float) * info.n_pad, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dx, x, sizeof(float) * M * C, hipMemcpyHostToDevice));

  /* ---- the operator */
  cf_conv_args a;
  memset(&a, 0, sizeof a);
  a.src[0] = dx; a.src_c[0] = C; a.n_src = 1;
  a.B = B; a.H = H; a.W = W; a.Ho = H; a.Wo = W; a.stride = 1;
  a.weight = dw; a.slots = ds; a.bias = db;
  a.K_pad = info.k_pad; a.N = N; a.N_pad = info.n_pad;
  a.out = dout; a.out_stride = N; a.out_layout = CF_LAYOUT_NHWC; a.act = CF_ACT_RELU;
  a.out_scale = info.out_scale;
  hipStream_t st;
  CHECK_HIP(hipStreamCreate(&st));
  CHECK_CF(cf_conv3x3_f16x3(&a, st));
  CHECK_HIP(hipStreamSynchronize(st));
  float* out = malloc(sizeof(float) * M * N);
  CHECK_HIP(hipMemcpy(out, dout, sizeof(float) * M * N, hipMemcpyDeviceToHost));

  /* ---- reference: conv -> BatchNorm (eval) -> ReLU in double, as the reference's modules compute it */
  double max_ref = 0.0, max_err = 0.0;
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < H; ++y)
      for (int xx = 0; xx < W; ++xx)
        for (int n = 0; n < N; ++n) {
          double acc = 0.0;
          for (int r = 0; r < 3; ++r)
            for (int q = 0; q < 3; ++q) {
              const int yy = y + r - 1, xq = xx + q - 1;
              if (yy < 0 || yy >= H || xq < 0 || xq >= W) continue;
              const float* px = x + (((long)b * H + yy) * W + xq) * C;
              for (int c = 0; c < C; ++c) acc += (double)px[c] * (double)w[((n * C + c) * 3 + r) * 3 + q];
            }
          double v = (acc - mean[n]) / sqrt((double)var[n] + 1e-5) * gamma[n] + beta[n];
          if (v < 0.0) v = 0.0;
          const double e = fabs(v - (double)out[(((long)b * H + y) * W + xx) * N + n]);
          if (v > max_ref) max_ref = v;
          if (e > max_err) max_err = e;
        }
  printf("cf_abi_version %d, packed %zu weight bytes, %d slots; max |err| / max |ref| = %.3e\n", cf_abi_version(), info.weight_bytes,
         info.n_slots, max_err / max_ref);
  return max_err / max_ref < 2e-6 ? 0 : 1;
}
