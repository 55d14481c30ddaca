// One 32-channel block of one pixel of the mx feature rows (layout: head_patch16_kernel<.., MX> in cf_heads.hip), shared by
// cf_pack_feat_mx (cf_heads.hip) and the epilogue of cf_dcn_v2_f16x3 (cf_gemm_f16.hip): v = clamp(scale x) (scale: a power of two, 16 by default), hi = fp16(v),
// lo = v - hi (exact); block exponent = smallest e with max|.| <= 7.5 * 2^e, from the bits of the maximum (exponent field - 2,
// + 1 if the mantissa exceeds 1.875); fields by v_cvt_scalef32_pk32_fp6_f16 / v_cvt_scalef32_2xpk16_fp6_f32 (value / scale, RNE,
// saturating; field j = channel j).
#pragma once
#include "cf_common.h"

namespace {

typedef _Float16 mxh16x32 __attribute__((ext_vector_type(32)));
typedef int mxi32x6 __attribute__((ext_vector_type(6)));
typedef unsigned int mxu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int mxu32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int mx_block_exp(float amax) {   // amax >= 0
  const int bits = __builtin_bit_cast(int, amax);
  const int e = ((bits >> 23) & 0xff) - 127 - 2 + ((bits & 0x7fffff) > 0x700000 ? 1 : 0);
  return e < -127 ? -127 : e;      // (amax = 0, fp32 denormals and maxima below 2^-125: the E8M0 code stays >= 0 - never 255 = NaN)
}

// x: the 32 channels [32 blk, 32 blk + 32) of one pixel (already activated), row: the pixel's 272-byte row
__device__ __forceinline__ void mx_pack_block(const float (&x)[32], unsigned char* row, int blk, float scale) {
  mxh16x32 h;
  f32x16 le, lo_;                              // lo: even / odd channels (the f32 convert interleaves its two sources)
  float mh = 0.0f, ml = 0.0f;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const float v = __builtin_amdgcn_fmed3f(x[j] * scale, -65504.0f, 65504.0f);
    const _Float16 hj = (_Float16)v;
    const float l = v - (float)hj;
    h[j] = hj;
    if (j & 1) lo_[j >> 1] = l; else le[j >> 1] = l;
    mh = fmaxf(mh, fabsf((float)hj));
    ml = fmaxf(ml, fabsf(l));
  }
  const int eh = mx_block_exp(mh), el = mx_block_exp(ml);
  // (a block of zeros - or of values below 2^-125, whose exponent is clamped - converts with scale 1: every field is then 0)
  const float sh = eh <= -127 ? 1.0f : __builtin_bit_cast(float, (eh + 127) << 23);
  const float sl = el <= -127 ? 1.0f : __builtin_bit_cast(float, (el + 127) << 23);
  // inline asm with an early-clobber destination: hipcc (ROCm 7.2) may allocate the 6-dword result ON TOP of the scale
  // (or a source) register of the builtin form, and the instruction writes its result in passes while still reading them
  // - every field behind the first pair then converts with a clobbered scale (seen: v_cvt_... v[0:5], .., .., v0)
  mxi32x6 h6, l6;
  asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(h6) : "v"(h), "v"(sh));
  asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(l6) : "v"(le), "v"(lo_), "v"(sl));
  {
    const mxu32x4* hs = reinterpret_cast<const mxu32x4*>(&h);
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<mxu32x4*>(row + 64 * i + 16 * blk) = hs[i];     // channels 32 blk + 8 i ..: segment i
  }
  *reinterpret_cast<mxu32x4*>(row + 64 * blk + 32) = mxu32x4{(unsigned)l6[0], (unsigned)l6[1], (unsigned)l6[2], (unsigned)l6[3]};
  *reinterpret_cast<mxu32x4*>(row + 64 * blk + 48) = mxu32x4{(unsigned)l6[4], (unsigned)l6[5], 0u, 0u};
  *reinterpret_cast<mxu32x4*>(row + 64 * (2 + blk) + 32) = mxu32x4{(unsigned)h6[0], (unsigned)h6[1], (unsigned)h6[2], (unsigned)h6[3]};
  *reinterpret_cast<mxu32x4*>(row + 64 * (2 + blk) + 48) = mxu32x4{(unsigned)h6[4], (unsigned)h6[5], 0u, 0u};
  row[256 + blk] = (unsigned char)(el + 127);
  row[258 + blk] = (unsigned char)(eh + 127);
  if (blk == 0) {
    *reinterpret_cast<unsigned*>(row + 260) = 0u;
    *reinterpret_cast<mxu32x2*>(row + 264) = mxu32x2{0u, 0u};
  }
}

}  // namespace
