// Shared pieces of the f16x3 kernels (cf_gemm_f16.hip, cf_conv3x3_f16.hip): operand split, weight
// fragment addressing.  See the numerics note at the top of cf_gemm_f16.hip.
#pragma once
#include "cf_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr float ASCALE = 16.0f;   // activation pre-scale (2^4), undone by out_scale
constexpr int FROWB = 80;         // LDS bytes per pixel row per plane: 32 f16 + 16 B pad

__device__ __forceinline__ unsigned pack_h2(_Float16 a, _Float16 b) {
  return ((unsigned)__builtin_bit_cast(unsigned short, b) << 16) | __builtin_bit_cast(unsigned short, a);
}

// 8 fp32 -> scaled, clamped, split into fp16 hi / lo (4 dwords each)
__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, u32x4& hi, u32x4& lo) {
  _Float16 h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float x = (e < 4 ? v0[e] : v1[e - 4]) * ASCALE;
    x = fminf(fmaxf(x, -65504.0f), 65504.0f);
    h[e] = (_Float16)x;
    l[e] = (_Float16)(x - (float)h[e]);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = pack_h2(h[2 * e], h[2 * e + 1]);
    lo[e] = pack_h2(l[2 * e], l[2 * e + 1]);
  }
}

__device__ __forceinline__ const f16x8* wfrag16(const unsigned char* w, int rt, int ks, int plane, int n_ks, int lane) {
  return reinterpret_cast<const f16x8*>(w + ((((size_t)rt * n_ks + ks) * 2 + plane) * 64 + lane) * 16);
}

}  // namespace
