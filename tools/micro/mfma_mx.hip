// Dev microbenchmark / layout probe for the heads' "f16 main + block-scaled FP6 cross terms" scheme (gate b).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_mx mfma_mx.hip && ./mfma_mx
// Part 1 (exact data, must print OK): the operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 with FP6 e2m3 operands -
//   lane l = 16 g + i holds row / column i, K block g (K = 32 g .. 32 g + 31), element j in bits [6j, 6j + 6) of its 6 dwords
//   (little endian), one E8M0 scale byte per lane (2^(b - 127)) selected by op_sel; C/D as every 16x16 MFMA - and of the two
//   FP6 pack-converts (value / scale, RNE, saturating; pk32_f16: field j = element j; 2xpk16_f32: field 2i = src0[i], 2i+1 = src1[i]).
// Part 2: rate of one 128-deep K slab of a 64 x 128 wave tile, every operand re-read from LDS:
//   bf16x3 (shipped):  12 x v_mfma_f32_16x16x32_bf16 per 16x16 tile
//   f16 + fp6:          4 x v_mfma_f32_16x16x32_f16 + 2 x v_mfma_scale_f32_16x16x128_f8f6f4 (cbsz = blgp = 2) per tile
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

// ---------------------------------------------------------------- part 1: layout
template <int OPA, int OPB>
__global__ void probe_mfma(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x4* c) {
  const int l = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 2, 2, OPA, sa[l], OPB, sb[l]);
  c[l] = acc;
}
__global__ void probe_cvt_f32(const f32x16* x, i32x6* o, float s) {
  const int l = threadIdx.x;
  o[l] = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(x[2 * l], x[2 * l + 1], s);
}
__global__ void probe_cvt_f16(const f16x32* x, i32x6* o, float s) {
  const int l = threadIdx.x;
  o[l] = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(x[l], s);
}

static float e2m3_value(int code) {
  const int s = code >> 5, e = (code >> 3) & 3, m = code & 7;
  const float v = e == 0 ? m / 8.0f : ldexpf(1.0f + m / 8.0f, e - 1);
  return s ? -v : v;
}
static void put6(unsigned* dw, int j, int code) {            // element j -> bits [6j, 6j+6) little endian over 6 dwords
  const int bit = 6 * j;
  unsigned long long v = (unsigned long long)(code & 63) << (bit & 31);
  dw[bit >> 5] |= (unsigned)v;
  if ((bit & 31) > 26) dw[(bit >> 5) + 1] |= (unsigned)(v >> 32);
}
static int get6(const unsigned* dw, int j) {
  const int bit = 6 * j;
  unsigned long long v = dw[bit >> 5];
  if ((bit >> 5) + 1 < 6) v |= (unsigned long long)dw[(bit >> 5) + 1] << 32;
  return (int)((v >> (bit & 31)) & 63);
}
static float e2m3_rne(float t) {                                // reference rounding: nearest, ties to even code, saturating
  float best = 0.0f; double bd = 1e30; int bc = 0;
  for (int c = 0; c < 32; ++c) {
    const double d = fabs(fabs((double)t) - e2m3_value(c));
    if (d < bd || (d == bd && (c & 1) == 0 && (bc & 1) == 1)) { bd = d; best = e2m3_value(c); bc = c; }
  }
  return t < 0 ? -best : best;
}

static int part1() {
  int bad = 0;
  unsigned *da, *db; int *dsa, *dsb; float* dc;
  CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dc, 64 * 16));
  for (int trial = 0; trial < 8; ++trial) {
    // A[16][128], B[128][16] as e2m3 codes; scale bytes per (row, K block) in byte `op` of the lane's scale dword
    static int A[16][128], B[128][16], SA[16][4], SB[16][4];
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) { A[i][k] = rand() & 63; B[k][i] = rand() & 63; }
    for (int i = 0; i < 16; ++i) for (int g = 0; g < 4; ++g) { SA[i][g] = 120 + rand() % 12; SB[i][g] = 121 + rand() % 12; }
    const int opa = trial & 3, opb = (trial >> 1) & 3;
    std::vector<unsigned> ha(64 * 8, 0u), hb(64 * 8, 0u);
    std::vector<int> hsa(64), hsb(64);
    for (int l = 0; l < 64; ++l) {
      const int i = l & 15, g = l >> 4;
      for (int j = 0; j < 32; ++j) { put6(&ha[l * 8], j, A[i][32 * g + j]); put6(&hb[l * 8], j, B[32 * g + j][i]); }
      ha[l * 8 + 6] = ha[l * 8 + 7] = 0xdeadbeefu;               // must be ignored for FP6
      hb[l * 8 + 6] = hb[l * 8 + 7] = 0xdeadbeefu;
      unsigned junk = (unsigned)rand() * 2654435761u;
      hsa[l] = (int)((junk & ~(0xffu << (8 * opa))) | ((unsigned)SA[i][g] << (8 * opa)));
      junk = (unsigned)rand() * 2246822519u;
      hsb[l] = (int)((junk & ~(0xffu << (8 * opb))) | ((unsigned)SB[i][g] << (8 * opb)));
    }
    CK(hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice));
#define LAUNCH(OA, OB) hipLaunchKernelGGL((probe_mfma<OA, OB>), dim3(1), dim3(64), 0, 0, (const i32x8*)da, (const i32x8*)db, dsa, dsb, (f32x4*)dc)
    switch (opa * 4 + opb) {
      case 0: LAUNCH(0, 0); break; case 1: LAUNCH(0, 1); break; case 2: LAUNCH(0, 2); break; case 3: LAUNCH(0, 3); break;
      case 4: LAUNCH(1, 0); break; case 5: LAUNCH(1, 1); break; case 6: LAUNCH(1, 2); break; case 7: LAUNCH(1, 3); break;
      case 8: LAUNCH(2, 0); break; case 9: LAUNCH(2, 1); break; case 10: LAUNCH(2, 2); break; case 11: LAUNCH(2, 3); break;
      case 12: LAUNCH(3, 0); break; case 13: LAUNCH(3, 1); break; case 14: LAUNCH(3, 2); break; default: LAUNCH(3, 3); break;
    }
    float hc[64 * 4];
    CK(hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost));
    int tb = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r, col = l & 15;         // C/D of every 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg
        double ref = 0.0;
        for (int k = 0; k < 128; ++k)
          ref += (double)e2m3_value(A[row][k]) * ldexp(1.0, SA[row][k >> 5] - 127) * e2m3_value(B[k][col]) * ldexp(1.0, SB[col][k >> 5] - 127);
        if (fabs(ref - hc[l * 4 + r]) > 1e-5 * (1.0 + fabs(ref))) { if (++tb <= 3) printf("  mismatch lane %d reg %d: got %g want %g\n", l, r, hc[l * 4 + r], ref); }
      }
    printf("[layout] scaled MFMA trial %d (op_sel a=%d b=%d): %s\n", trial, opa, opb, tb ? "MISMATCH" : "OK");
    bad += tb;
  }
  // pack-converts: 32 f32 (two 16-vectors) / 32 f16 -> 6 dwords, value / scale rounded to e2m3
  {
    std::vector<float> hx(64 * 32); std::vector<_Float16> hh(64 * 32);
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
      float v = (rand() / (float)RAND_MAX - 0.5f) * 20.0f;      // includes saturation beyond 7.5 * scale
      if (l == 0) v = (j - 16) * 0.0625f * 4.0f;                // exact ties on the grid (scale 4)
      hh[l * 32 + j] = (_Float16)v; hx[l * 32 + j] = (float)hh[l * 32 + j];
    }
    float* dx; _Float16* dh; unsigned* dout;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dh, hh.size() * 2)); CK(hipMalloc(&dout, 64 * 32));   // an i32x6 is stored with a 32-byte stride
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dh, hh.data(), hh.size() * 2, hipMemcpyHostToDevice));
    const float scale = 4.0f;
    for (int which = 0; which < 2; ++which) {
      if (which == 0) hipLaunchKernelGGL(probe_cvt_f32, dim3(1), dim3(64), 0, 0, (const f32x16*)dx, (i32x6*)dout, scale);
      else hipLaunchKernelGGL(probe_cvt_f16, dim3(1), dim3(64), 0, 0, (const f16x32*)dh, (i32x6*)dout, scale);
      std::vector<unsigned> ho(64 * 8);
      CK(hipMemcpy(ho.data(), dout, 64 * 32, hipMemcpyDeviceToHost));
      // hypotheses for the field order of the f32 form: (0) field j = element j of [src0 | src1]; (1) interleaved 2i = src0[i], 2i+1 = src1[i]
      for (int hyp = (which == 0 ? 1 : 0); hyp < (which == 0 ? 2 : 1); ++hyp) {   // f32 form: fields interleave the two sources
        int tb = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
          const int src = hyp == 0 ? j : ((j & 1) * 16 + (j >> 1));
          const float want = e2m3_rne(hx[l * 32 + src] / scale);
          const float got = e2m3_value(get6(&ho[l * 8], j));
          if (want != got && !(want == 0.0f && got == 0.0f)) { if (++tb <= 3) printf("  cvt lane %d field %d: x %g got %g want %g\n", l, j, hx[l * 32 + src], got, want); }
        }
        printf("[layout] %s field-order hypothesis %d (value / scale, RNE, saturating): %s\n", which == 0 ? "cvt_scalef32_2xpk16_fp6_f32" : "cvt_scalef32_pk32_fp6_f16", hyp, tb ? "MISMATCH" : "OK");
        bad += tb;
      }
    }
  }
  return bad;
}

// ---------------------------------------------------------------- part 2: rate
constexpr int LDS_FRAGS = 40;   // 16-byte-per-lane fragments resident in LDS (40 KiB)
template <int MODE>
__global__ __launch_bounds__(256, 2) void rate(const i32x4* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ i32x4 lds[64 * LDS_FRAGS];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 64 * LDS_FRAGS; i += 256) lds[i] = src[(blockIdx.x % 64) * 64 * LDS_FRAGS + i];
  __syncthreads();
  f32x4 acc[4][8];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 8; ++b) for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
  auto frag = [&](int n) { return lds[(n % LDS_FRAGS) * 64 + lane]; };
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {               // bf16x3: per 32-deep k-step 4 + 4 A fragments, 8 + 8 B fragments, 96 MFMAs
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 Ah[4], Al[4], Bh[8], Bl[8];
#pragma unroll
        for (int a = 0; a < 4; ++a) { Ah[a] = __builtin_bit_cast(bf16x8, frag(it + ks * 7 + a)); Al[a] = __builtin_bit_cast(bf16x8, frag(it + ks * 7 + 4 + a)); }
#pragma unroll
        for (int b = 0; b < 8; ++b) { Bh[b] = __builtin_bit_cast(bf16x8, frag(it + ks * 5 + 8 + b)); Bl[b] = __builtin_bit_cast(bf16x8, frag(it + ks * 5 + 16 + b)); }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 8; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al[a], Bh[b], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 8; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[a], Bl[b], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 8; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[a], Bh[b], acc[a][b], 0, 0, 0);
      }
    } else {                        // f16 main: 4 k-steps x 32 MFMAs; cross: 2 scaled MFMAs per tile (K = [hi6 ; lo6] of 128 channels)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        f16x8 A[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) A[a] = __builtin_bit_cast(f16x8, frag(it + ks * 7 + a));
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          f16x8 B[4];
#pragma unroll
          for (int b = 0; b < 4; ++b) B[b] = __builtin_bit_cast(f16x8, frag(it + ks * 5 + 8 + 4 * hf + b));
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][4 * hf + b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[a], B[b], acc[a][4 * hf + b], 0, 0, 0);
        }
      }
#pragma unroll
      for (int cs = 0; cs < 2; ++cs) {
        i32x8 A6[4], B6[8];
        int sa, sb[2];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const i32x4 lo = frag(it + cs * 11 + a), hi = frag(it + cs * 11 + 4 + a);      // 24 of these 32 bytes are used
          A6[a] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], 0, 0};
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const i32x4 lo = frag(it + cs * 13 + 8 + b), hi = frag(it + cs * 13 + 16 + b);
          B6[b] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], 0, 0};
        }
        { const i32x4 s = frag(it + cs * 3 + 30); sa = (s[0] & 0x03030303) + 0x7c7c7c7c; sb[0] = (s[1] & 0x03030303) + 0x7c7c7c7c; sb[1] = (s[2] & 0x03030303) + 0x7c7c7c7c; }
#define MX(a, b, oa, ob) acc[a][b] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A6[a], B6[b], acc[a][b], 2, 2, oa, sa, ob, sb[(b) >> 2])
#define MXROW(a, oa) MX(a, 0, oa, 0); MX(a, 1, oa, 1); MX(a, 2, oa, 2); MX(a, 3, oa, 3); MX(a, 4, oa, 0); MX(a, 5, oa, 1); MX(a, 6, oa, 2); MX(a, 7, oa, 3)
        MXROW(0, 0); MXROW(1, 1); MXROW(2, 2); MXROW(3, 3);
      }
    }
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 8; ++b) for (int r = 0; r < 4; ++r) s += acc[a][b][r];
  out[blockIdx.x * 256 + tid] = s;
}

int main() {
  const int bad = part1();
  printf("[layout] %s\n", bad ? "FAILED" : "all scaled-MFMA layout checks OK");
  const int blocks = 512 * 8, iters = 400;
  std::vector<_Float16> h((size_t)64 * 64 * LDS_FRAGS * 8);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.25f);
  i32x4* d; float* o;
  CK(hipMalloc(&d, h.size() * 2)); CK(hipMalloc(&o, blocks * 256 * 4));
  CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // model FLOPs of a 128-deep slab of a 64 x 128 wave tile, 4 waves: the same product either way
  const double flop = (double)blocks * 4 * iters * 64.0 * 128 * 128 * 2;
  double ms_mode[2] = {0, 0};
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 2; ++mode) {
      for (int w = 0; w < 3; ++w) {
        if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
        else hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
      }
      hipEventRecord(e0);
      const int n = 5;
      for (int w = 0; w < n; ++w) {
        if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
        else hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      ms_mode[mode] = ms / n;
      printf("[rate] %s: %.3f ms per launch = %.0f model TFLOP/s\n", mode == 0 ? "bf16x3 (12 MFMA / tile / 128 K)" : "f16 + fp6 (4 + 2 MFMA / tile / 128 K)", ms / n, flop / (ms / n * 1e-3) / 1e12);
    }
  printf("[rate] speed-up f16+fp6 over bf16x3: %.2fx (gate: >= 1.4x)\n", ms_mode[0] / ms_mode[1]);
  return bad ? 1 : 0;
}
