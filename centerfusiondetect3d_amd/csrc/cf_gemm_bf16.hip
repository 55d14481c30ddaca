// Implicit-GEMM convolution on the gfx950 bf16 MFMA pipe with SPLIT operands ("bf16x3").
//
// Every fp32 value x is carried as two bf16 numbers x = hi + lo (hi = rne(x), lo = rne(x - hi),
// 16 significant bits together) and a product is evaluated as
//        a*b  ~=  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi          (fp32 accumulate, a_lo*b_lo dropped)
// i.e. three v_mfma_f32_32x32x16_bf16 per 16-deep k-step.  Relative error per product <= ~2^-17,
// 5.3x the fp32-MFMA rate (3 x 32 cycles per 16 k instead of 8 x 64).  Used for the heads only:
// they sit behind the DCN neck, so their rounding is not amplified (DESIGN.md §4 "Numerics").
//
// Layout ("split-bf16 NHWC"): a pixel is [C hi][C lo] bf16, i.e. 4*C bytes - the same HBM bytes as
// fp32.  One K-slot = 8 consecutive channels of one source at one tap = one 16-byte unit per plane.
// Weights: [N_pad][2][K_pad] bf16.  GEMM/tiling/LDS-padding logic mirrors cf_gemm.hip: BM x 32 A
// chunk and BN x 32 weight chunk per step, rows padded to 80 bytes per plane (conflict-free
// ds_read_b128), 4 waves x (TM x TN) 32x32 accumulators, next chunk prefetched to registers.
#include <stdlib.h>
#include "cf_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));


struct ConvB {
  const unsigned char* src[CF_MAX_SRC];
  int src_c[CF_MAX_SRC];
  const unsigned char* weight;
  const cf_slot* slots;
  const float* bias;
  unsigned char* out;  // split-bf16 NHWC or fp32 NCHW
  float* out2;
  int H, W, Ho, Wo, stride, K_pad, n_chunks, NT;
  int out_stride, out_layout, act, M, N, HoWo;
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const __bf16 x = (__bf16)a, y = (__bf16)b;
  return ((unsigned)__builtin_bit_cast(unsigned short, y) << 16) | __builtin_bit_cast(unsigned short, x);
}
__device__ __forceinline__ float bf16_round(float a) { return (float)(__bf16)a; }

__device__ __forceinline__ float act_f(float v, int act) {
  if (act == CF_ACT_RELU) return fmaxf(v, 0.0f);
  if (act == CF_ACT_SIGMOID_CLAMP) return fminf(fmaxf(cf_sigmoid(v), 1e-4f), 1.0f - 1e-4f);
  return v;
}

#ifdef CF_LEGACY_HEADS   // (the unfused heads' bf16x3 convolution of rounds 1-2: nothing dispatches it in the default build)
// BK = K depth staged per main-loop step (32 or 64 bf16).  LDS rows are BK*2 + 16 bytes per plane:
// an odd number of 16-byte slots, so the 16 rows a ds_read_b128 lane group touches never collide.
template <int BM, int BN, int WAVES_M, int WAVES_N, int BK>
__global__ __launch_bounds__(256) void conv_bf16x3_kernel(ConvB p) {
  constexpr int ROWB = BK * 2 + 16;
  constexpr int UPR = BK / 8;        // 16-byte units per row per plane
  constexpr int UPT = UPR / 4;       // units each staging thread moves per row (1 or 2)
  constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
  constexpr int RA = BM / 32, RB = BN / 32;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  __shared__ __attribute__((aligned(16))) unsigned char smem[(BM + BN) * 2 * ROWB];
  // planes: A_hi | A_lo | B_hi | B_lo
  constexpr int A_LO = BM * ROWB, B_HI = 2 * BM * ROWB, B_LO = 2 * BM * ROWB + BN * ROWB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lid = cf_xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid / p.NT, nt = lid - mt * p.NT;
  const int m0 = mt * BM, n0 = nt * BN;
  const int tr = tid >> 3, ts = tid & 7;
  const int plane = ts >> 2, unit = ts & 3;

  int y0[RA], x0[RA], boff[RA];
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int m = m0 + tr + 32 * j;
    if (m < p.M) {
      const int b = m / p.HoWo, rem = m - b * p.HoWo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      y0[j] = ho * p.stride;
      x0[j] = wo * p.stride;
      boff[j] = b * p.H * p.W;
    } else {
      y0[j] = -(1 << 28);
      x0[j] = 0;
      boff[j] = 0;
    }
  }

  u32x4 ra[UPT][RA], rb[UPT][RB];
  auto load_chunk = [&](int c) {
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int un = unit + 4 * u;                       // unit inside the BK-wide row
      const int sidx = c * UPR + un;                     // slot index (8 channels per slot)
      const cf_slot sl = p.slots[sidx];
      // a 32-wide half-chunk never mixes sources (host packing), units un and un+4 may
      const int src = __builtin_amdgcn_readfirstlane(p.slots[c * UPR + 4 * u].src);
      const unsigned char* sp = src == 1 ? p.src[1] : src == 2 ? p.src[2] : src == 3 ? p.src[3] : p.src[0];
      const int sc = src == 1 ? p.src_c[1] : src == 2 ? p.src_c[2] : src == 3 ? p.src_c[3] : p.src_c[0];
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        const int y = y0[j] + sl.dy, x = x0[j] + sl.dx;
        const bool ok = (src >= 0) && (sl.c_off >= 0) && ((unsigned)y < (unsigned)p.H) &&
                        ((unsigned)x < (unsigned)p.W);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ok)
          v = *reinterpret_cast<const u32x4*>(sp + ((size_t)(boff[j] + y * p.W + x) * (2 * sc) + plane * sc + sl.c_off) * 2);
        ra[u][j] = v;
      }
      const unsigned char* wp = p.weight + (((size_t)(n0 + tr) * 2 + plane) * p.K_pad + c * BK + un * 8) * 2;
#pragma unroll
      for (int j = 0; j < RB; ++j) rb[u][j] = *reinterpret_cast<const u32x4*>(wp + (size_t)(32 * j) * 2 * p.K_pad * 2);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int li = lane & 31, h = lane >> 5;
  load_chunk(0);
  for (int c = 0; c < p.n_chunks; ++c) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
#pragma unroll
      for (int j = 0; j < RA; ++j)
        *reinterpret_cast<u32x4*>(smem + plane * A_LO + (tr + 32 * j) * ROWB + (unit + 4 * u) * 16) = ra[u][j];
#pragma unroll
      for (int j = 0; j < RB; ++j)
        *reinterpret_cast<u32x4*>(smem + B_HI + plane * (B_LO - B_HI) + (tr + 32 * j) * ROWB + (unit + 4 * u) * 16) = rb[u][j];
    }
    __syncthreads();
    if (c + 1 < p.n_chunks) load_chunk(c + 1);
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
      const int koff = s * 32 + h * 16;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int row = (wm * TM + tm) * 32 + li;
        ah[tm] = *reinterpret_cast<const bf16x8*>(smem + row * ROWB + koff);
        al[tm] = *reinterpret_cast<const bf16x8*>(smem + A_LO + row * ROWB + koff);
      }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int row = (wn * TN + tn) * 32 + li;
        bh[tn] = *reinterpret_cast<const bf16x8*>(smem + B_HI + row * ROWB + koff);
        bl[tn] = *reinterpret_cast<const bf16x8*>(smem + B_LO + row * ROWB + koff);
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
        }
    }
  }

  // ---- epilogue.  C/D layout: col = lane&31 (channel n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel)
  const int m_base = m0 + wm * TM * 32, n_base = n0 + wn * TN * 32;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n_base + tn * 32 + li;
    const bool n_ok = n < p.N;
    const float bias = n_ok ? p.bias[n] : 0.0f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int m4 = m_base + tm * 32 + 8 * g + 4 * h;
        if (p.out_layout == CF_LAYOUT_NHWC_SPLIT_BF16) {
          // two lanes (n even / n odd) swap one value each so that every lane owns a (n, n+1)
          // pair of one pixel row: 4-byte stores instead of 2-byte ones
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const float v0 = act_f(acc[tm][tn][g * 4 + e] + bias, p.act);
            const float v1 = act_f(acc[tm][tn][g * 4 + e + 1] + bias, p.act);
            const float send = (li & 1) ? v0 : v1;
            const float recv = __shfl_xor(send, 1);
            const float lo_c = (li & 1) ? recv : v0;   // value of the even column
            const float hi_c = (li & 1) ? v1 : recv;   // value of the odd column
            const int m = m4 + e + (li & 1);
            const int n2 = n & ~1;
            if (m < p.M && n2 + 1 < p.N + (p.N & 1)) {
              const float a_hi = bf16_round(lo_c), b_hi = bf16_round(hi_c);
              unsigned char* o = p.out + ((size_t)m * 2 * p.out_stride + n2) * 2;
              *reinterpret_cast<unsigned*>(o) = pack_bf16(a_hi, b_hi);
              *reinterpret_cast<unsigned*>(o + (size_t)p.out_stride * 2) = pack_bf16(lo_c - a_hi, hi_c - b_hi);
            }
          }
        } else {  // fp32 NCHW
          if (n_ok && m4 < p.M) {
            float* outf = reinterpret_cast<float*>(p.out);
            const int b = m4 / p.HoWo, pix = m4 - b * p.HoWo;
            const size_t o = ((size_t)b * p.N + n) * p.HoWo + pix;
            f32x4 v, v2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float raw = acc[tm][tn][g * 4 + e] + bias;
              v[e] = act_f(raw, p.act);
              v2[e] = 0.0f;
              if (p.act == CF_ACT_RAW_AND_SIGDEPTH) v2[e] = 1.0f / (cf_sigmoid(raw) + 1e-6f) - 1.0f;
            }
            if (m4 + 3 < p.M && pix + 3 < p.HoWo && (o & 3) == 0) {
              *reinterpret_cast<f32x4*>(&outf[o]) = v;
              if (p.act == CF_ACT_RAW_AND_SIGDEPTH) *reinterpret_cast<f32x4*>(&p.out2[o]) = v2;
            } else {
              for (int e = 0; e < 4; ++e) {
                const int m = m4 + e;
                if (m < p.M) {
                  const int bb = m / p.HoWo, pp = m - bb * p.HoWo;
                  const size_t oo = ((size_t)bb * p.N + n) * p.HoWo + pp;
                  outf[oo] = v[e];
                  if (p.act == CF_ACT_RAW_AND_SIGDEPTH) p.out2[oo] = v2[e];
                }
              }
            }
          }
        }
      }
    }
  }
}

#endif  // CF_LEGACY_HEADS

// fp32 NHWC [M][C] -> split-bf16 [M][2][Cs] (Cs >= C, channels C..Cs-1 zero), 8 channels per thread
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, unsigned char* __restrict__ out,
                                                         long M, int C, int in_stride, int Cs) {
  const int G = Cs / 8;
  const long total = M * G;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / G;
    const int c0 = (int)(i - m * G) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (c0 + e < C) ? x[m * in_stride + c0 + e] : 0.0f;
    u32x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = bf16_round(v[2 * e]), b = bf16_round(v[2 * e + 1]);
      hi[e] = pack_bf16(a, b);
      lo[e] = pack_bf16(v[2 * e] - a, v[2 * e + 1] - b);
    }
    unsigned char* o = out + ((size_t)m * 2 * Cs + c0) * 2;
    *reinterpret_cast<u32x4*>(o) = hi;
    *reinterpret_cast<u32x4*>(o + (size_t)Cs * 2) = lo;
  }
}

}  // namespace

extern "C" int cf_conv2d_bf16x3(const cf_conv_args* a, void* stream) {
  CF_REQUIRE(a != nullptr, "cf_conv2d_bf16x3: null args");
#ifndef CF_LEGACY_HEADS
  // the unfused heads' convolution (rounds 1-2): the fused head launches (cf_head_fused) carry every bf16x3 layer of the path now
  (void)stream;
  CF_REQUIRE(false, "cf_conv2d_bf16x3 is a legacy kernel path: rebuild libcfhip with -DCF_LEGACY_HEADS (CF_EXTRA_FLAGS); the heads "
                    "run on cf_head_fused, fp32-accurate convolutions on cf_conv2d_f16x3 / cf_conv2d_fused");
  return CF_EINVAL;
#else
  CF_REQUIRE(a->n_src >= 1 && a->n_src <= CF_MAX_SRC, "cf_conv2d_bf16x3: n_src=%d", a->n_src);
  CF_REQUIRE(a->K_pad > 0 && a->K_pad % 32 == 0, "cf_conv2d_bf16x3: K_pad=%d not a multiple of 32", a->K_pad);
  CF_REQUIRE(a->N_pad >= a->N && a->N_pad % 32 == 0 && a->N > 0, "cf_conv2d_bf16x3: N=%d N_pad=%d", a->N, a->N_pad);
  CF_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->Ho > 0 && a->Wo > 0 && a->stride > 0, "cf_conv2d_bf16x3: bad geometry");
  CF_REQUIRE(a->weight && a->slots && a->bias && a->out, "cf_conv2d_bf16x3: null buffer");
  CF_REQUIRE(a->residual == nullptr, "cf_conv2d_bf16x3: residual is not supported");
  CF_REQUIRE(a->out_layout == CF_LAYOUT_NCHW || a->out_layout == CF_LAYOUT_NHWC_SPLIT_BF16,
             "cf_conv2d_bf16x3: out_layout must be NCHW (fp32) or NHWC_SPLIT_BF16");
  CF_REQUIRE(a->out_layout == CF_LAYOUT_NCHW || (a->out_stride >= a->N && a->out_stride % 2 == 0 && a->N % 2 == 0),
             "cf_conv2d_bf16x3: split output needs even N and out_stride >= N");
  CF_REQUIRE(a->act != CF_ACT_RAW_AND_SIGDEPTH || (a->out2 && a->out_layout == CF_LAYOUT_NCHW),
             "cf_conv2d_bf16x3: RAW_AND_SIGDEPTH needs out2 and NCHW");
  for (int i = 0; i < a->n_src; ++i)
    CF_REQUIRE(a->src[i] && a->src_c[i] > 0 && a->src_c[i] % 8 == 0, "cf_conv2d_bf16x3: source %d invalid", i);
  const long M = (long)a->B * a->Ho * a->Wo;
  CF_REQUIRE((long)a->B * a->H * a->W < (1L << 30) && M < (1L << 31), "cf_conv2d_bf16x3: tensor too large");
  ConvB k{};
  for (int i = 0; i < CF_MAX_SRC; ++i) {
    k.src[i] = i < a->n_src ? reinterpret_cast<const unsigned char*>(a->src[i]) : nullptr;
    k.src_c[i] = i < a->n_src ? a->src_c[i] : 0;
  }
  k.weight = reinterpret_cast<const unsigned char*>(a->weight);
  k.slots = a->slots;
  k.bias = a->bias;
  k.out = reinterpret_cast<unsigned char*>(a->out);
  k.out2 = a->out2;
  k.H = a->H; k.W = a->W; k.Ho = a->Ho; k.Wo = a->Wo; k.stride = a->stride;
  k.K_pad = a->K_pad;
  k.out_stride = a->out_stride; k.out_layout = a->out_layout; k.act = a->act;
  k.M = (int)M; k.N = a->N; k.HoWo = a->Ho * a->Wo;
  hipStream_t st = (hipStream_t)stream;
  const int MT = (int)((M + 127) / 128);
  // BK = 64 halves the barriers but measured 8 % slower than BK = 32 on the 3x3 head layers
  // (2 workgroups per CU either way, longer exposed prologue): opt-in for experiments only.
  static const bool want64 = getenv("CF_BF16_BK64") != nullptr;
  const bool bk64 = (a->K_pad % 64 == 0) && want64;
  k.n_chunks = a->K_pad / (bk64 ? 64 : 32);
  if (a->N_pad % 128 == 0) {
    k.NT = a->N_pad / 128;
    if (bk64) {
      static std::once_flag once;
      std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16x3_kernel<128, 128, 2, 2, 64>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 0);
      });
      hipLaunchKernelGGL((conv_bf16x3_kernel<128, 128, 2, 2, 64>), dim3(MT * k.NT), dim3(256), 0, st, k);
    } else {
      hipLaunchKernelGGL((conv_bf16x3_kernel<128, 128, 2, 2, 32>), dim3(MT * k.NT), dim3(256), 0, st, k);
    }
  } else if (a->N_pad % 64 == 0) {
    k.NT = a->N_pad / 64;
    k.n_chunks = a->K_pad / 32;
    hipLaunchKernelGGL((conv_bf16x3_kernel<128, 64, 2, 2, 32>), dim3(MT * k.NT), dim3(256), 0, st, k);
  } else {
    k.NT = a->N_pad / 32;
    k.n_chunks = a->K_pad / 32;
    hipLaunchKernelGGL((conv_bf16x3_kernel<128, 32, 4, 1, 32>), dim3(MT * k.NT), dim3(256), 0, st, k);
  }
  return cf_check_launch("cf_conv2d_bf16x3");
#endif
}

extern "C" int cf_split_bf16(const float* x, void* out, long M, int C, int in_stride, int Cs, void* stream) {
  CF_REQUIRE(x && out, "cf_split_bf16: null buffer");
  CF_REQUIRE(M > 0 && C > 0 && in_stride >= C && Cs >= C && Cs % 8 == 0, "cf_split_bf16: bad geometry");
  const long total = M * (Cs / 8);
  long g = (total + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x,
                     reinterpret_cast<unsigned char*>(out), M, C, in_stride, Cs);
  return cf_check_launch("cf_split_bf16");
}
