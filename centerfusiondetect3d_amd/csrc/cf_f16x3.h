// Shared pieces of the f16x3 kernels (cf_gemm_f16.hip, cf_conv3x3_f16.hip): operand split, weight
// fragment addressing.  See the numerics note at the top of cf_gemm_f16.hip.
#pragma once
#include "cf_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr float ASCALE = 16.0f;   // DEFAULT activation pre-scale (2^4; the kernels take theirs from the argument block: in_scale), undone by out_scale
constexpr int FROWB = 80;         // LDS bytes per pixel row per plane: 32 f16 + 16 B pad

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_h2(_Float16 a, _Float16 b) {
  return ((unsigned)__builtin_bit_cast(unsigned short, b) << 16) | __builtin_bit_cast(unsigned short, a);
}

// two ALREADY SCALED fp32 values -> clamped, split into packed fp16 hi / lo pairs.  Written on
// 2-vectors so gfx950's packed converts are used (v_cvt_pk_f16_f32, v_pk_add_f32): 4 VALU
// instructions per value including the clamp, and no separate packing.
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
  const f32x2 x = {__builtin_amdgcn_fmed3f(a, -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(b, -65504.0f, 65504.0f)};
  const f16x2 h = __builtin_convertvector(x, f16x2);
  const f16x2 l = __builtin_convertvector(x - __builtin_convertvector(h, f32x2), f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

// 8 fp32 -> scaled by the layer's activation pre-scale, clamped, split into fp16 hi / lo (4 dwords each)
__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, u32x4& hi, u32x4& lo, float in_scale) {
  const f32x4 a = v0 * in_scale, b = v1 * in_scale;
  { unsigned th, tl; split2(a[0], a[1], th, tl); hi[0] = th; lo[0] = tl; }
  { unsigned th, tl; split2(a[2], a[3], th, tl); hi[1] = th; lo[1] = tl; }
  { unsigned th, tl; split2(b[0], b[1], th, tl); hi[2] = th; lo[2] = tl; }
  { unsigned th, tl; split2(b[2], b[3], th, tl); hi[3] = th; lo[3] = tl; }
}

// Weight fragment of (32-row tile rt, k-step ks, plane): [rt][ks][plane][64 lanes][8 f16].  Written as (tile base) +
// 16 * lane so that with a wave-uniform rt / ks (callers read the wave index through readfirstlane) the base is scalar
// arithmetic and ONE per-lane 32-bit offset register serves every weight load of a kernel.
__device__ __forceinline__ const f16x8* wfrag16(const unsigned char* w, int rt, int ks, int plane, int n_ks, int lane) {
  const unsigned char* base = w + ((size_t)rt * n_ks + ks) * 2048 + plane * 1024;
  return reinterpret_cast<const f16x8*>(base + (unsigned)lane * 16u);
}

}  // namespace
