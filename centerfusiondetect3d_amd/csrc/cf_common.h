// Shared device/host helpers for libcfhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include "cf_hip.h"

// Dev timing arms and probes (-DCF_DCN_NOLOAD, -DCF_CONV3_PROF, ...: kernels that skip a part of their work to time the rest -
// their results are garbage by design - or that write timestamps into their output).  They exist only in builds that ALSO
// define CF_DEV_ARMS (the tools/ab_*.sh / profile_*.sh scripts do; build.py never does): without it every arm macro is
// undefined right here, so no stray -D flag can put an arm into the product library, and this list is the complete
// inventory of what a reader of the kernels may skip.
#ifndef CF_DEV_ARMS
#undef CF_CONV3_NOBARRIER
#undef CF_CONV3_NOPATCH
#undef CF_CONV3_NOSTAGE
#undef CF_CONV3_NOWEIGHT
#undef CF_CONV3_PROF
#undef CF_CONV3_SHAPE16T
#undef CF_DCN_DEEP1
#undef CF_DCN_NOBARRIER
#undef CF_DCN_NOBLEND
#undef CF_DCN_NODESC
#undef CF_DCN_NOEPI
#undef CF_DCN_NOGATHER
#undef CF_DCN_NOLOAD
#undef CF_DCN_NOMFMA
#undef CF_DCN_NOPIN
#undef CF_DCN_NOWEIGHT
#undef CF_DCN_PROF
#undef CF_DCN_SMALLTILE
#undef CF_MX_ARM_NOA
#undef CF_MX_ARM_NOB
#undef CF_MX_ARM_NOCVT
#undef CF_MX_ARM_NORED
#undef CF_NO_WAVE_SYNC
#undef CF_ONESET
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CF_BK 32          // K chunk (floats) staged per main-loop step
#define CF_LDS_STRIDE 36  // BK + 4: ds_read_b128 of 16 rows hits 16 distinct 4-bank slots

void cf_set_error(const char* fmt, ...);

#define CF_REQUIRE(cond, ...)    \
  do {                           \
    if (!(cond)) {               \
      cf_set_error(__VA_ARGS__); \
      return CF_EINVAL;          \
    }                            \
  } while (0)

// Dynamic-LDS limit of one kernel: raised (never lowered) with hipFuncSetAttribute the first time a launch needs
// more than what has been set.  One static instance per kernel / template instantiation; safe when several host
// threads issue launches of the same kernel concurrently (the C ABI is re-entrant per stream).
struct CfLdsLimit {
  std::atomic<size_t> limit{0};
  std::mutex m;
  template <typename K>
  void ensure(K kernel, size_t dyn, size_t floor_bytes) {
    if (dyn <= limit.load(std::memory_order_acquire) && limit.load(std::memory_order_relaxed) > 0) return;
    std::lock_guard<std::mutex> lk(m);
    const size_t cur = limit.load(std::memory_order_relaxed);
    if (cur > 0 && dyn <= cur) return;
    const size_t v = dyn < floor_bytes ? floor_bytes : dyn;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)v);
    limit.store(v, std::memory_order_release);
  }
};

static inline int cf_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    cf_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return CF_ELAUNCH;
  }
  return CF_OK;
}

// Bijective remap of the linear workgroup id so that consecutive logical tiles run on ONE XCD
// (hardware deals workgroups round-robin over the 8 XCDs; each XCD has its own L2).
__device__ __forceinline__ int cf_xcd_remap(int b, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Lanes of ONE wave exchanging data through LDS (write as (row, channel), read back as (pixel, chunk)): orders this wave's
// earlier LDS accesses before its later ones for the compiler (no reordering across it) and, through the wavefront-scope
// release / acquire pair, in the memory model; the hardware executes one wave's DS operations in issue order anyway.
__device__ __forceinline__ void cf_wave_lds_sync() {
#ifndef CF_NO_WAVE_SYNC   // (dev A/B only: -DCF_NO_WAVE_SYNC measures what the ordering costs - nothing)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (workgroup scope: emits s_waitcnt lgkmcnt(0))
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// Activation pre-scale of the f16x3 / mx kernels as it arrives over the ABI (cf_conv_args.in_scale, ...): 0 = the default 16;
// anything else has to be a positive, finite power of two (so that scaling and un-scaling are exact).  -> the scale, or 0 if invalid.
static inline float cf_resolve_in_scale(float s) {
  if (s == 0.0f) return 16.0f;
  uint32_t b;
  memcpy(&b, &s, 4);
  const uint32_t e = (b >> 23) & 0xff;
  return ((b >> 31) == 0 && (b & 0x7fffff) == 0 && e > 0 && e < 255) ? s : 0.0f;
}

__device__ __forceinline__ float cf_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
