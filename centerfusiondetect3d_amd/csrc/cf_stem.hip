// DLA-34 stem in ONE launch:  images (NCHW, 3 ch)  ->  base_layer 7x7 (3 -> 16) + BN + ReLU
//                                                   ->  level0 3x3 (16 -> 16) + BN + ReLU
//                                                   ->  level1 3x3 stride 2 (16 -> 32) + BN + ReLU  (fp32 NHWC)
// (model/networks/dla.py:237-262: base_layer, level0, level1 of DLA.forward).  As separate launches
// these three layers move 1.7 GB of full-resolution activations through HBM for 66 GFLOP; fused, a
// workgroup keeps everything between the image and the half-resolution level1 map in LDS (52 KB):
//
//   level1 tile 8 x 8  <-  level0 region 17 x 17  <-  base region 19 x 19  <-  image patch 25 x 25
//
// Every region position outside the image is written as ZERO (each layer pads its OWN input with
// zeros - evaluating the previous layer outside the image would be wrong).
//
// Arithmetic: f16x3 as in cf_gemm_f16.hip (fp32 values split into fp16 hi + lo after a power-of-two
// scale, products on the f16 MFMA pipe, fp32 accumulation), here on v_mfma_f32_16x16x32_f16: 16 output
// channels are exactly one tile row, pixels are the columns (lane & 15), and a lane's 8 k-values
// (lane >> 4 selects the k group) are
//   * base_layer: ONE tap of the image patch, [4 ch hi | 4 ch lo] - a single 16-byte LDS read.  Both
//     halves meet the weights as  {w_hi, w_hi} . {x_hi, x_lo}  +  {w_lo, 0} . {x_hi, x_lo}, i.e. the same
//     three products  w_hi x_hi + w_hi x_lo + w_lo x_hi  in two MFMAs (13 k-steps of 4 taps; 49 taps + 3 pad);
//   * level0 / level1: 8 channels of one tap (k-step = 2 taps x 16 ch, 5 k-steps, tap 9 is padding),
//     hi and lo planes read separately, three MFMAs (main + two cross terms) as everywhere else.
// Weight fragments are packed on the host in exactly that lane order (packing.pack_stem).
#include "cf_f16x3.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int ST_T1 = 8;                       // level1 tile edge
constexpr int ST_R0 = 2 * ST_T1 + 1;           // level0 region edge (17)
constexpr int ST_RB = ST_R0 + 2;               // base region edge (19)
constexpr int ST_RI = ST_RB + 6;               // image patch edge (25)
constexpr int ST_ROWB = 80;                    // bytes per region pixel: 16 ch hi (32) + lo (32) + pad
constexpr int ST_IN_B = ST_RI * ST_RI * 16;    // image patch: [px][4 hi | 4 lo]
constexpr int ST_BASE_B = ST_RB * ST_RB * ST_ROWB;
constexpr int ST_L0_B = ST_R0 * ST_R0 * ST_ROWB;
constexpr int ST_LDS = ST_BASE_B + ST_L0_B;    // 28,880 + 23,120 = 52,000 B: three workgroups per CU (the image
                                               // patch, dead once the base region exists, shares the level0 area)
static_assert(ST_IN_B <= ST_L0_B, "the image patch must fit the level0 area");

struct StemK {
  const float* x;                 // NCHW fp32 (B, C, H, W), C <= 3 read
  int B, C, H, W;
  const unsigned char* w_base;    // [13 ks][2: {hi,hi} / {lo,0}][64 lanes][8 f16]
  const unsigned char* w_l0;      // [5 ks][2: hi / lo][64][8]
  const unsigned char* w_l1;      // [2 rt][5 ks][2][64][8]
  const float* b_base;            // 16
  const float* b_l0;              // 16
  const float* b_l1;              // 32
  float s_base, s_l0, s_l1;       // 2^-s / in_scale per layer
  float a_img, a_base, a_l0;      // activation pre-scales (in_scale): image, base_layer output, level0 output
  float* out;                     // fp32 NHWC (B, H/2, W/2, 32)
  float* out_pool;                // optional: its 2x2 / stride 2 max-pool, fp32 NHWC (B, H/4, W/4, 32)
  int tiles_x, tiles_y;
};

__device__ __forceinline__ const f16x8* sfrag(const unsigned char* w, int idx, int lane) {
  return reinterpret_cast<const f16x8*>(w + ((size_t)idx * 64 + lane) * 16);
}

// 4 fp32 (one pixel, 4 consecutive channels) -> scaled, clamped fp16 hi / lo pairs
__device__ __forceinline__ void split4(const f32x4v& v, uint2& hi, uint2& lo, float in_scale) {
  const f32x4v xs = v * in_scale;
  split2(xs[0], xs[1], hi.x, lo.x);
  split2(xs[2], xs[3], hi.y, lo.y);
}

__global__ __launch_bounds__(256, 3) void stem_kernel(StemK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* base_lds = lds;
  unsigned char* l0_lds = lds + ST_BASE_B;
  unsigned char* in_lds = l0_lds;            // P0/P1 only; P2 starts behind a barrier

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, kg = lane >> 4;
  const int per_img = p.tiles_x * p.tiles_y;
  const int b = blockIdx.x / per_img, rem = blockIdx.x - b * per_img;
  const int oy0 = (rem / p.tiles_x) * ST_T1, ox0 = (rem % p.tiles_x) * ST_T1;   // level1 tile origin
  const int y_l0 = 2 * oy0 - 1, x_l0 = 2 * ox0 - 1;          // level0 region origin (full resolution)
  const int y_b = y_l0 - 1, x_b = x_l0 - 1;                  // base region origin
  const int y_i = y_b - 3, x_i = x_b - 3;                    // image patch origin
  const long HW = (long)p.H * p.W;
  // workgroups whose whole image patch lies inside the image (almost all of them) skip every border test
  const bool interior = y_i >= 0 && x_i >= 0 && y_i + ST_RI <= p.H && x_i + ST_RI <= p.W;

  // ---- P0: image patch -> split fp16 -> LDS (zeros outside the image, channel 3 is zero); every load
  //      of the thread is in flight before the first one is used
  {
    constexpr int NQ = (ST_RI * ST_RI + 255) / 256;
    f32x4v v[NQ];
#pragma unroll
    for (int it = 0; it < NQ; ++it) {
      const int q = tid + 256 * it;
      const int y = y_i + q / ST_RI, x = x_i + q % ST_RI;
      v[it] = f32x4v{0.f, 0.f, 0.f, 0.f};
      if (q < ST_RI * ST_RI && (interior || ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W))) {
        const float* src = p.x + (size_t)b * p.C * HW + (size_t)y * p.W + x;
        v[it][0] = src[0];
        if (p.C > 1) v[it][1] = src[HW];
        if (p.C > 2) v[it][2] = src[2 * HW];
      }
    }
#pragma unroll
    for (int it = 0; it < NQ; ++it) {
      const int q = tid + 256 * it;
      uint2 hi, lo;
      split4(v[it], hi, lo, p.a_img);
      if (q < ST_RI * ST_RI) *reinterpret_cast<u32x4*>(in_lds + q * 16) = u32x4{hi.x, hi.y, lo.x, lo.y};
    }
  }

  // ---- P1 weights (both variants of all 13 k-steps stay in registers) and per-lane tap offsets
  f16x8 wb[13][2];
  int toff[13];
#pragma unroll
  for (int ks = 0; ks < 13; ++ks) {
    wb[ks][0] = *sfrag(p.w_base, ks * 2 + 0, lane);
    wb[ks][1] = *sfrag(p.w_base, ks * 2 + 1, lane);
    // taps 49..51 are padding (zero weights).  Four compile-time candidates, one select per k group:
    // no per-lane division chain
    auto off = [](int tap) { tap = tap < 48 ? tap : 48; return ((tap / 7) * ST_RI + tap % 7) * 16; };
    const int o01 = kg & 1 ? off(4 * ks + 1) : off(4 * ks + 0), o23 = kg & 1 ? off(4 * ks + 3) : off(4 * ks + 2);
    toff[ks] = kg & 2 ? o23 : o01;
  }
  const f32x4v bias_b = *reinterpret_cast<const f32x4v*>(p.b_base + 4 * kg);
  __syncthreads();

  // ---- P1: base_layer over the 19 x 19 region, 16 pixels per MFMA tile, two tiles in flight per wave
  //      (independent accumulators keep the MFMA pipe busy between dependent steps)
  constexpr int NB = ST_RB * ST_RB;                          // 361
  constexpr int NTB = (NB + 15) / 16;                        // 23
  for (int t0 = wave; t0 < NTB; t0 += 8) {
    int q[2], py[2], px[2];
    const unsigned char* src[2];
    f32x4v acc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      q[u] = min((t0 + 4 * u) * 16 + col, NB - 1);
      py[u] = q[u] / ST_RB;
      px[u] = q[u] - py[u] * ST_RB;
      src[u] = in_lds + (py[u] * ST_RI + px[u]) * 16;
      acc[u] = f32x4v{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < 13; ++ks) {
      f16x8 xv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) xv[u] = *reinterpret_cast<const f16x8*>(src[u] + toff[ks]);
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ks][1], xv[u], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ks][0], xv[u], acc[u], 0, 0, 0);
    }
    // lane = (pixel col, channel group kg): channels 4kg..4kg+3 of pixel q
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int y = y_b + py[u], x = x_b + px[u];
      const bool inside = interior || ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W);
      f32x4v v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = inside ? fmaxf(acc[u][e] * p.s_base + bias_b[e], 0.0f) : 0.0f;
      uint2 hi, lo;
      split4(v, hi, lo, p.a_base);
      if ((t0 + 4 * u) * 16 + col < NB) {
        *reinterpret_cast<uint2*>(base_lds + q[u] * ST_ROWB + 8 * kg) = hi;
        *reinterpret_cast<uint2*>(base_lds + q[u] * ST_ROWB + 32 + 8 * kg) = lo;
      }
    }
  }

  // ---- P2 weights / offsets: k-step = taps 2ks, 2ks+1; lane k group: tap 2ks + (kg >> 1), channels 8(kg & 1)..+8
  f16x8 w0h[5], w0l[5];
  int t0off[5];
#pragma unroll
  for (int ks = 0; ks < 5; ++ks) {
    w0h[ks] = *sfrag(p.w_l0, ks * 2 + 0, lane);
    w0l[ks] = *sfrag(p.w_l0, ks * 2 + 1, lane);
    auto off = [](int tap) { tap = tap < 8 ? tap : 8; return ((tap / 3) * ST_RB + tap % 3) * ST_ROWB; };
    t0off[ks] = (kg & 2 ? off(2 * ks + 1) : off(2 * ks)) + (kg & 1) * 16;
  }
  const f32x4v bias_0 = *reinterpret_cast<const f32x4v*>(p.b_l0 + 4 * kg);
  __syncthreads();

  // ---- P2: level0 over the 17 x 17 region, two tiles in flight per wave
  constexpr int N0 = ST_R0 * ST_R0;                          // 289
  constexpr int NT0 = (N0 + 15) / 16;                        // 19
  for (int t0 = wave; t0 < NT0; t0 += 8) {
    int q[2], py[2], px[2];
    const unsigned char* src[2];
    f32x4v accm[2], accs[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      q[u] = min((t0 + 4 * u) * 16 + col, N0 - 1);
      py[u] = q[u] / ST_R0;
      px[u] = q[u] - py[u] * ST_R0;
      src[u] = base_lds + (py[u] * ST_RB + px[u]) * ST_ROWB;
      accm[u] = f32x4v{0.f, 0.f, 0.f, 0.f};
      accs[u] = accm[u];
    }
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      f16x8 xh[2], xl[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        xh[u] = *reinterpret_cast<const f16x8*>(src[u] + t0off[ks]);
        xl[u] = *reinterpret_cast<const f16x8*>(src[u] + t0off[ks] + 32);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) accs[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0l[ks], xh[u], accs[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) accm[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0h[ks], xh[u], accm[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) accs[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0h[ks], xl[u], accs[u], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int y = y_l0 + py[u], x = x_l0 + px[u];
      const bool inside = interior || ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W);
      f32x4v v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = inside ? fmaxf((accm[u][e] + accs[u][e]) * p.s_l0 + bias_0[e], 0.0f) : 0.0f;
      uint2 hi, lo;
      split4(v, hi, lo, p.a_l0);
      if ((t0 + 4 * u) * 16 + col < N0) {
        *reinterpret_cast<uint2*>(l0_lds + q[u] * ST_ROWB + 8 * kg) = hi;
        *reinterpret_cast<uint2*>(l0_lds + q[u] * ST_ROWB + 32 + 8 * kg) = lo;
      }
    }
  }

  // ---- P3 weights / offsets (stride 2: out (oy, ox) reads level0 region (2oy + ky, 2ox + kx)).
  //      Wave w owns the 16-channel half rt = w >> 1 of pixels 32 (w & 1) .. +32: one half's weights per wave
  const int rt = wave >> 1;
  f16x8 w1h[5], w1l[5];
  int t1off[5];
#pragma unroll
  for (int ks = 0; ks < 5; ++ks) {
    w1h[ks] = *sfrag(p.w_l1, (rt * 5 + ks) * 2 + 0, lane);
    w1l[ks] = *sfrag(p.w_l1, (rt * 5 + ks) * 2 + 1, lane);
    auto off = [](int tap) { tap = tap < 8 ? tap : 8; return ((tap / 3) * ST_R0 + tap % 3) * ST_ROWB; };
    t1off[ks] = (kg & 2 ? off(2 * ks + 1) : off(2 * ks)) + (kg & 1) * 16;
  }
  const f32x4v bias_1 = *reinterpret_cast<const f32x4v*>(p.b_l1 + 16 * rt + 4 * kg);
  __syncthreads();

  // ---- P3: level1, 64 output pixels x 32 channels
  {
    const unsigned char* src[2];
    f32x4v accm[2], accs[2];
    int oy[2], ox[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = (wave & 1) * 32 + u * 16 + col;
      oy[u] = q >> 3;
      ox[u] = q & 7;
      src[u] = l0_lds + ((2 * oy[u]) * ST_R0 + 2 * ox[u]) * ST_ROWB;
      accm[u] = f32x4v{0.f, 0.f, 0.f, 0.f};
      accs[u] = accm[u];
    }
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      f16x8 xh[2], xl[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        xh[u] = *reinterpret_cast<const f16x8*>(src[u] + t1off[ks]);
        xl[u] = *reinterpret_cast<const f16x8*>(src[u] + t1off[ks] + 32);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) accs[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1l[ks], xh[u], accs[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) accm[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h[ks], xh[u], accm[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) accs[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h[ks], xl[u], accs[u], 0, 0, 0);
    }
    const int H1 = p.H / 2, W1 = p.W / 2;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int y = oy0 + oy[u], x = ox0 + ox[u];
      f32x4v v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf((accm[u][e] + accs[u][e]) * p.s_l1 + bias_1[e], 0.0f);
      if (y < H1 && x < W1) *reinterpret_cast<f32x4v*>(p.out + (((size_t)b * H1 + y) * W1 + x) * 32 + 16 * rt + 4 * kg) = v;
      // the level-2 Tree max-pools this map 2x2 (dla.py:96 downsample) - the only reader of that pool is its `project`: the
      // 16 pixels of the MFMA tile are two rows of 8, so a pool window is lanes {c, c + 1, c + 8, c + 9} of one k group:
      // two lane exchanges, and the even-column lanes of the upper row write the pooled pixel (floor semantics at odd sizes)
      if (p.out_pool) {
        f32x4v m = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          m[e] = fmaxf(m[e], __shfl_xor(m[e], 1));
          m[e] = fmaxf(m[e], __shfl_xor(m[e], 8));
        }
        const int H2 = H1 / 2, W2 = W1 / 2, yp = y >> 1, xp = x >> 1;
        if ((col & 9) == 0 && yp < H2 && xp < W2)
          *reinterpret_cast<f32x4v*>(p.out_pool + (((size_t)b * H2 + yp) * W2 + xp) * 32 + 16 * rt + 4 * kg) = m;
      }
    }
  }
}

}  // namespace

extern "C" int cf_stem_fused(const cf_stem_args* a, void* stream) {
  CF_REQUIRE(a != nullptr, "cf_stem_fused: null args");
  CF_REQUIRE(a->x && a->out, "cf_stem_fused: null tensor");
  CF_REQUIRE(a->B > 0 && a->C >= 1 && a->C <= 3, "cf_stem_fused: B=%d C=%d", a->B, a->C);
  CF_REQUIRE(a->H > 0 && a->W > 0 && a->H % 2 == 0 && a->W % 2 == 0, "cf_stem_fused: H=%d W=%d must be even", a->H, a->W);
  CF_REQUIRE(a->w_base && a->w_level0 && a->w_level1 && a->b_base && a->b_level0 && a->b_level1, "cf_stem_fused: null weights");
  CF_REQUIRE(a->scale_base > 0.f && a->scale_level0 > 0.f && a->scale_level1 > 0.f, "cf_stem_fused: out scales missing");
  StemK k{};
  k.x = a->x; k.B = a->B; k.C = a->C; k.H = a->H; k.W = a->W;
  k.w_base = reinterpret_cast<const unsigned char*>(a->w_base);
  k.w_l0 = reinterpret_cast<const unsigned char*>(a->w_level0);
  k.w_l1 = reinterpret_cast<const unsigned char*>(a->w_level1);
  k.b_base = a->b_base; k.b_l0 = a->b_level0; k.b_l1 = a->b_level1;
  k.s_base = a->scale_base; k.s_l0 = a->scale_level0; k.s_l1 = a->scale_level1;
  k.a_img = cf_resolve_in_scale(a->in_scale[0]); k.a_base = cf_resolve_in_scale(a->in_scale[1]); k.a_l0 = cf_resolve_in_scale(a->in_scale[2]);
  CF_REQUIRE(k.a_img > 0.f && k.a_base > 0.f && k.a_l0 > 0.f, "cf_stem_fused: in_scale must be 0 (= 16) or a power of two");
  k.out = a->out;
  k.out_pool = a->out_pool;
  k.tiles_x = (a->W / 2 + ST_T1 - 1) / ST_T1;
  k.tiles_y = (a->H / 2 + ST_T1 - 1) / ST_T1;
  const long blocks = (long)k.tiles_x * k.tiles_y * a->B;
  CF_REQUIRE(blocks < (1L << 31) && (long)a->B * a->C * a->H * a->W < (1L << 40), "cf_stem_fused: tensor too large");
  static CfLdsLimit lds_limit;
  lds_limit.ensure(stem_kernel, ST_LDS, ST_LDS);
  hipLaunchKernelGGL(stem_kernel, dim3((unsigned)blocks), dim3(256), ST_LDS, (hipStream_t)stream, k);
  return cf_check_launch("cf_stem_fused");
}
