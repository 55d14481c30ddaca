// 3x3 / stride 1 / pad 1 convolution on the f16x3 scheme (fp32 storage, split-fp16 products - see the
// numerics note in cf_gemm_f16.hip) with PATCH REUSE: the generic slot kernel re-gathers and re-splits
// every input element once per tap (9x); here a workgroup owns R = 64*WP consecutive pixels of the
// flattened (B*H*W) index space and, per 16-channel slice, stages the flat row range
//     [m0 - W - 1, m0 + R + W + 1)
// ONCE - fp32 rows -> fp16 hi/lo -> LDS.  All 9 taps then read their B fragments from that patch at
// row offset (dy+1)*W + (dx+1); a lane whose tap leaves the image (or the batch element) is pointed
// at an all-zero row instead, so borders cost one select per fragment address.  Per slice that is one
// barrier and one operand split for 9 k-steps of MFMA work, and (R + 2W + 2)/R <= 2.6x input
// traffic instead of 9x.
//
// K order is SLICE-MAJOR: k-step index ks = slice * 9 + tap (packing.pack_conv_f16(slice_major=True)
// emits the slot table in that order, so the generic kernel cf_conv2d_f16x3 computes the same sums
// from the same packed weights - it is the fallback for feature maps too wide for the LDS patch).
//
// Workgroup = 4 waves as WC (32*RT-channel groups) x WP (64-pixel groups) x WK (slices in flight:
// the patch then carries 16*WK channels per round and wave wk multiplies slice wk; partial sums are
// added in fixed wave order through LDS at the end).  WK > 1 is for small feature maps with many
// input channels (the 14x25 / 28x50 offset convolutions), where pixel tiles alone cannot fill 256 CUs.
//
// STRIDE 2 (S2, tiled form only; the four BasicBlock conv1 layers that open levels 2-5, dla.py:124-145): an output tile of
// TH x 16 pixels needs the (2 TH + 1) x 33 input pixels around it.  The patch keeps each input row as two PLANES - the
// 17 odd columns 2 (x0 + j) - 1, then the 16 even columns 2 (x0 + j) - so that tap (dy, dx) of output pixel (py, px) sits
// at row 2 py + dy, entry {px, 17 + px, px + 1}[dx]: a compile-time offset per tap, exactly as in the stride-1 tile.  Every
// input element is fetched and split ~1.1 times instead of the slot kernel's 2.25.
//
// Replaces the 3x3 convolutions of model/networks/dla.py:42-62, 124-145 (BasicBlock conv1/conv2) and
// the conv_offset_mask of model/networks/dla.py:406-414 (DeformConv) on the device.
#include <stdlib.h>
#include "cf_f16x3.h"

namespace {

#ifdef CF_CONV3_SHAPE16T
// (dev timing arm, results are garbage: every v_mfma_f32_32x32x16_f16 becomes two v_mfma_f32_16x16x32_f16 on quarters of
//  the same accumulator - the same FLOPs, LDS reads, weight loads and registers, only the MFMA shape differs.  Costs the
//  16x16x32 rewrite of this kernel before building it: DESIGN.md section 9.)
__device__ __forceinline__ f32x16 mfma_shape16(f16x8 a, f16x8 b, f32x16 c) {
  f32x4 q0 = {c[0], c[1], c[2], c[3]}, q1 = {c[8], c[9], c[10], c[11]};
  q0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, q0, 0, 0, 0);
  q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, q1, 0, 0, 0);
  c[0] = q0[0]; c[1] = q0[1]; c[2] = q0[2]; c[3] = q0[3];
  c[8] = q1[0]; c[9] = q1[1]; c[10] = q1[2]; c[11] = q1[3];
  return c;
}
#define CF_MFMA_F16(a, b, c) mfma_shape16(a, b, c)
#else
#define CF_MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif

struct Conv3F {
  const float* x;               // fp32 NHWC, first channel of the source
  const unsigned char* weight;  // [N_pad/32][n_ks][2][64][8 f16]
  const float* bias;
  const float* residual;
  float* out;
  int x_stride, H, W, HW, M, N, n_rt, n_ks, n_rounds, res_stride, out_stride, act, PR;   // H, W, HW, M: the OUTPUT map
  int tiles_x, tiles_y;         // T2 only
  int Hi, Wi, HWi;              // S2 only: the input map (H, W are then the output's)
  // ROOT only: the Tree's 1x1 Root over (this convolution's output x2, its residual x1) run from the epilogue
  const unsigned char* root_w;  // [2][8 k-steps][2][64][8 f16]: K = (x2 channels 0..63, x1 channels 0..63)
  const float* root_bias;
  float* root_out;
  int root_out_stride, root_act;
  float root_scale;
  // ... and the Root's further sources (the Tree's children: dla.py:109-117), K order behind x2 and x1
  const float* root_xsrc[2];
  int root_xsrc_c[2], root_xsrc_ch[2];   // floats per pixel, channels (multiples of 64)
  int root_nks;                          // k-steps of the whole Root: (2 N + children's channels) / 16
  float out_scale;
  float in_scale;        // activation pre-scale of this convolution's operands (patch rows, projected rows)
  float root_in_scale;   // ROOT only: ... and of the Root GEMM's operands (x2, x1, the children)
  // PROJ only: a 1x1 convolution of a second tensor (same map as the output) summed into the same accumulators
  const float* proj_x;
  int proj_c, proj_ch, proj_nks, proj_ks0;   // floats per pixel, channels (multiple of 32), k-steps, first k-step in the stream
};

// T2: the R pixels are an (R/16) x 16 tile of ONE image instead of a flat run: the patch is the tile
// plus a one-pixel frame, (R/16 + 2) x 18 rows, zero-filled outside the image - so no tap needs a
// validity select and the tap offsets are compile-time constants.  Pays on wide maps (W = 200: 1.3x
// input traffic instead of 2.6x) whenever W is close to a multiple of 16.
// CT = 32-pixel column tiles per wave: 2 (64 x 64 wave tiles, two waves per SIMD at 256 registers each) or 4 - the
// ONE-WAVE-PER-SIMD form (MINB = 1, 512 registers: both accumulator sets of a 64-channel x 128-pixel tile, 256 registers,
// sit in AGPRs; every weight fragment is fetched once per 128 pixels instead of once per 64).
// ROOT (BasicBlock conv2 of a one-level Tree without children, dla.py:105-118, 33-41; N = 64 WC channels, all of them in
// this workgroup): the Tree's Root - ReLU(W_root . [x2; x1] + b), x2 = this convolution's output, x1 = its residual -
// runs from the epilogue.  Every wave splits its 64 channels of x2 (bias, residual, ReLU applied) and of x1 to fp16
// hi / lo into its two LDS regions, which ARE pieces of the B operand of the 1x1 GEMM; after a barrier (WC > 1: the WC
// waves of a pixel group read each other's pieces) 8 WC k-steps of MFMAs follow in the slot kernel's order, each wave
// producing its own 64 output channels.  Only the Root's output goes to HBM: x2 is never written (unless p.out is
// given), x1 is read once for both uses, one launch less.
// PROJ (BasicBlock conv2 of the sub-tree that opens a DLA level: its residual is the Tree's `project` - 1x1 convolution
// + BN - of the 2x2-max-pooled level input, dla.py:96-107, 56-62): the projection's k-steps run behind the 3x3 part, into
// the same accumulators, so the residual never exists as a tensor and its launch is gone.  B tiles of 32 pixels x 64 channels:
// the WC x WK waves of a pixel group each fetch one 64-channel piece of the pooled rows from HBM (whole rows), split it
// to fp16 hi / lo into their LDS region, and after the group's barrier every wave multiplies the round's pieces in K
// order (WK > 1: piece i goes to K-split wave i % WK).  The regions are the patch buffers' memory (dead by then).
template <int WC, int WP, int WK, int RT, int NU, bool DB, int MINB, bool T2, int CT = 2, bool S2 = false, bool ROOT = false,
          bool PROJ = false>
__global__ __launch_bounds__(64 * WC * WP * WK, MINB) void conv3x3_f16x3_kernel(Conv3F p) {
  static_assert(!ROOT || (WK == 1 && RT == 2 && !S2 && 64 * WC * WP * WK == 256), "Root fusion: 64-channel waves, 4 waves, every channel of a pixel in the workgroup");
  static_assert(!PROJ || (!S2 && !ROOT), "projection k-steps: stride 1, no fused Root");
  constexpr int NT = 64 * WC * WP * WK;     // 4 waves, or 8 (WP doubled: two pixel groups share each weight fragment through L1)
  static_assert(NT == 256 || NT == 512, "4 or 8 waves per workgroup");
  static_assert(!S2 || (T2 && WK == 1), "stride 2: tiled form, no K split");
  constexpr int R = 32 * CT * WP;
  constexpr int PW2 = S2 ? 33 : 18;         // T2: entries per patch row (16 + 2; stride 2: 17 odd + 16 even input columns)
  constexpr int PYS = S2 ? 2 : 1;           // patch rows per output row
  constexpr int ROWB = 64 * WK + 16;        // per patch row: WK x (16 hi + 16 lo f16) + pad (odd multiple of 16 B)
  constexpr int UPR = 4 * WK;               // 16-byte fp32 units per patch row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  // the wave index is wave-uniform: read through readfirstlane so that everything derived from it (the wave's channel
  // group, its weight tile addresses, its k-steps) is scalar - SGPRs and scalar ALU instead of per-lane registers
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  const int wk = wave % WK, wp = (wave / WK) % WP, wc = wave / (WK * WP);
#ifdef CF_CONV3_PROF   // dev (tools/prof_conv3.py): cycles per phase of thread 0, written over its first output values
  long long t_prof[4] = {0, 0, 0, 0};
  long long t_last = clock64();
#define PROF_MARK(i) { const long long t_now = clock64(); t_prof[i] += t_now - t_last; t_last = t_now; }
#else
#define PROF_MARK(i)
#endif
  // consecutive tiles on ONE XCD (cf_xcd_remap): neighbouring patches overlap, and every round
  // re-touches the same rows - both should hit that XCD's L2
  const int bid = cf_xcd_remap(blockIdx.x, gridDim.x);
  int m0 = bid * R;                         // flat: first pixel; T2: first pixel of the image + tile origin below
  int ty0 = 0, tx0 = 0;
  if (T2) {
    const int per_img = p.tiles_x * p.tiles_y;
    const int b = bid / per_img, rem = bid - b * per_img;
    ty0 = (rem / p.tiles_x) * (R / 16);
    tx0 = (rem % p.tiles_x) * 16;
    m0 = b * p.HW;
  }
  const int m0i = S2 ? (m0 / p.HW) * p.HWi : m0;   // first pixel of the image in the INPUT map
  const int rt0 = (blockIdx.y * WC + wc) * RT;
  const bool w_ok = rt0 < p.n_rt;
  const int bufb = (p.PR + 1) * ROWB;       // + the zero row
  const int zrow = p.PR * ROWB + wk * 64 + h * 16;

  // zero rows of both buffers
  if (tid < ROWB / 4) {
    *reinterpret_cast<unsigned*>(smem + p.PR * ROWB + tid * 4) = 0u;
    if (DB) *reinterpret_cast<unsigned*>(smem + bufb + p.PR * ROWB + tid * 4) = 0u;
  }

  // ---- staging role: NU (patch row, 4-channel unit) pairs per thread
  int goff[NU];                              // element offset of the unit in x for round 0, -1 = zeros
#pragma unroll
  for (int it = 0; it < NU; ++it) {
    const int u = tid + NT * it;
    const int row = u / UPR, q = u % UPR;
    if (S2) {
      const int e = row % PW2;
      const int y = 2 * ty0 - 1 + row / PW2, x = e < 17 ? 2 * (tx0 + e) - 1 : 2 * (tx0 + e - 17);
      const bool ok = row < p.PR && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
      goff[it] = ok ? (m0i + y * p.Wi + x) * p.x_stride + 4 * q : -1;
    } else if (T2) {
      const int y = ty0 - 1 + row / PW2, x = tx0 - 1 + row % PW2;
      const bool ok = row < p.PR && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      goff[it] = ok ? (m0 + y * p.W + x) * p.x_stride + 4 * q : -1;
    } else {
      const int g = m0 - p.W - 1 + row;
      goff[it] = (row < p.PR && g >= 0 && g < p.M) ? g * p.x_stride + 4 * q : -1;
    }
  }
  // the next round's patch is staged in two halves (loads of the first half fly over taps 0-3, those of
  // the second over taps 4-8), so only half of the raw registers are live at any time.  The loads are
  // UNCONDITIONAL - a unit outside the image reads a valid dummy address and is zeroed when it is split - so no
  // exec-masked memory instruction sits inside the scheduling-pinned loop ("glitch")
  constexpr int NH0 = (NU + 1) / 2;
  f32x4 raw[NU];
  auto load_patch = [&](int r, int lo, int hi) {
    const int c0 = r * 16 * WK;
#ifdef CF_CONV3_NOPATCH   // (dev timing experiment: the patch is fetched once per tile, later rounds re-split the same rows)
    if (r > 0) return;
#endif
#pragma unroll
    for (int it = 0; it < NU; ++it) {
      if (it < lo || it >= hi) continue;
      raw[it] = *reinterpret_cast<const f32x4*>(p.x + max(goff[it], 0) + c0);
    }
  };
  auto store_patch = [&](unsigned char* buf, int lo, int hi) {
#ifdef CF_CONV3_NOSTAGE   // (dev timing experiment: no operand split / LDS store behind the first patch)
    if (buf != smem) return;
#endif
    int tid_s = threadIdx.x;                 // (laundered: the NU destination addresses are re-derived per call - 3 ALU
    asm volatile("" : "+v"(tid_s));          //  instructions each - instead of living in registers / scratch all loop long)
#pragma unroll
    for (int it = 0; it < NU; ++it) {
      if (it < lo || it >= hi) continue;
      const int u = tid_s + NT * it;
      const int row = u / UPR, q = u % UPR;
      if (row < p.PR) {
        const f32x4 xs = goff[it] >= 0 ? raw[it] * p.in_scale : f32x4{0.f, 0.f, 0.f, 0.f};   // (a select: the dummy read may hold anything)
        uint2 hi2, lo2;
        split2(xs[0], xs[1], hi2.x, lo2.x);
        split2(xs[2], xs[3], hi2.y, lo2.y);
        unsigned char* o = buf + row * ROWB + (q >> 2) * 64 + (q & 3) * 8;
        *reinterpret_cast<uint2*>(o) = hi2;
        *reinterpret_cast<uint2*>(o + 32) = lo2;
      }
    }
  };

  // ---- MFMA role: per column tile the patch row of this lane's pixel and its 9-bit tap validity
  int rowb[CT];
  unsigned vmask[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int pl = wp * (32 * CT) + ct * 32 + li;
    const int m = m0 + pl;
    rowb[ct] = (T2 ? (pl >> 4) * (PYS * PW2) + (pl & 15) : pl) * ROWB + wk * 64 + h * 16;
    unsigned mk = 0;
    if (T2) {
      mk = 0x1ffu;                           // the frame is part of the patch
    } else if (m < p.M) {
      const int rem = m % p.HW;
      const int y = rem / p.W, x = rem - y * p.W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) mk |= 1u << t;
      }
    }
    vmask[ct] = mk;
  }

  f32x16 accm[RT][CT], accs[RT][CT];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < CT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        accm[a][b][r] = 0.0f;
        accs[a][b][r] = 0.0f;
      }

  // weight fragments three taps ahead: set (t % 3) holds tap t
  f16x8 wh[3][RT], wl[3][RT];
  const int ks_last = (p.n_rounds * WK - WK + wk) * 9 + 8;      // this wave's last k-step
  // fragment address = scalar tile base (row tile, k-step: uniform) + 16 * lane + 1 KiB for the lo plane: one per-lane
  // 32-bit offset register serves every weight load of the kernel
  const unsigned lane16 = (unsigned)lane * 16u;
  auto load_w = [&](f16x8 (&dh)[RT], f16x8 (&dl)[RT], int ks) {
#ifdef CF_CONV3_NOWEIGHT  // (dev timing experiment: weight fragments fetched for the first three taps only)
    if (ks > wk * 9 + 2) return;
#endif
    ks = min(ks, ks_last);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const unsigned char* base = p.weight + ((size_t)(w_ok ? rt0 + rt : 0) * p.n_ks + ks) * 2048;
      dh[rt] = *reinterpret_cast<const f16x8*>(base + lane16);
      dl[rt] = *reinterpret_cast<const f16x8*>(base + 1024 + lane16);
    }
  };

  load_patch(0, 0, NU);
#pragma unroll
  for (int t = 0; t < 3; ++t) load_w(wh[t], wl[t], wk * 9 + t);
  store_patch(smem, 0, NU);
  __syncthreads();
  PROF_MARK(0)

  for (int r = 0; r < p.n_rounds; ++r) {
    const unsigned char* cur = smem + (DB ? (r & 1) * bufb : 0);
    unsigned char* nxt = smem + (DB ? ((r + 1) & 1) * bufb : 0);
    const bool more = r + 1 < p.n_rounds;
    if (more) load_patch(r + 1, 0, NH0);
    const int ks0 = (r * WK + wk) * 9;
    // B fragments: the hi plane one tap ahead (xh[t & 1]), the lo plane - needed only by the third
    // sweep - at the start of its tap; weights three taps ahead.  The sched_barrier after every tap
    // keeps the compiler from sinking those prefetches back down to their first use
    f16x8 xh[2][CT], xl[CT];
    auto x_addr = [&](int ct, int t, int toff) { return (T2 || ((vmask[ct] >> t) & 1u)) ? rowb[ct] + toff : zrow; };
    // stride 2: entry {px, 17 + px, px + 1} of patch row 2 py + dy (compile-time per tap)
    auto s2_off = [](int t) { return ((t / 3) * PW2 + (t % 3 == 0 ? 0 : t % 3 == 1 ? 17 : 1)) * ROWB; };
    int toff = 0;                            // ((t / 3) * W + t % 3) * ROWB, built incrementally
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) xh[0][ct] = *reinterpret_cast<const f16x8*>(cur + x_addr(ct, 0, 0));
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) xl[ct] = *reinterpret_cast<const f16x8*>(cur + x_addr(ct, t, toff) + 32);
      if (S2) toff = s2_off(t + 1);
      else toff += (t % 3 == 2) ? ((T2 ? PW2 : p.W) - 2) * ROWB : ROWB;
      if (t + 1 < 9) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          xh[(t + 1) & 1][ct] = *reinterpret_cast<const f16x8*>(cur + x_addr(ct, t + 1, toff));
      }
      // three independent sweeps over the tiles: no MFMA waits on the one issued just before it
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#ifdef CF_ONESET   // (dev timing experiment: all three products into ONE accumulator set - fails the float64 RMS gate)
          accm[rt][ct] = CF_MFMA_F16(wl[t % 3][rt], xh[t & 1][ct], accm[rt][ct]);
#else
          accs[rt][ct] = CF_MFMA_F16(wl[t % 3][rt], xh[t & 1][ct], accs[rt][ct]);
#endif
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          accm[rt][ct] = CF_MFMA_F16(wh[t % 3][rt], xh[t & 1][ct], accm[rt][ct]);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#ifdef CF_ONESET
          accm[rt][ct] = CF_MFMA_F16(wh[t % 3][rt], xl[ct], accm[rt][ct]);
#else
          accs[rt][ct] = CF_MFMA_F16(wh[t % 3][rt], xl[ct], accs[rt][ct]);
#endif
      // tap t+3 of this round, or tap t-6 of the next one (same set either way)
      load_w(wh[t % 3], wl[t % 3], t + 3 < 9 ? ks0 + t + 3 : ks0 + 9 * WK + t - 6);
      if (DB && t == 3 && more) {            // first half of the next patch: split + store, then request the rest
        store_patch(nxt, 0, NH0);
        load_patch(r + 1, NH0, NU);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    PROF_MARK(1)
    if (!DB) {
      __syncthreads();                       // single buffer: everyone is done reading it
      if (more) {
        store_patch(nxt, 0, NH0);
        load_patch(r + 1, NH0, NU);
      }
    }
    if (more) store_patch(nxt, NH0, NU);
#ifndef CF_CONV3_NOBARRIER   // (dev timing experiment: results are garbage without it)
    __syncthreads();
#endif
    PROF_MARK(2)
  }

  // ---- PROJ: the projection's k-steps, behind the 3x3 part (the patch is dead: its memory holds the B tiles)
  if constexpr (PROJ) {
    constexpr int BROW = 144;                // B tile: 64 f16 + 16 B per pixel row and plane
    constexpr int BPLANE = 32 * BROW;
    constexpr int REG = 2 * BPLANE;          // hi and lo plane of one piece: 9216 B
    constexpr int NG = WC * WK;              // waves of a pixel group
    int tid_p = threadIdx.x;                 // (every lane-derived index re-derived here, as in the epilogues: nothing of
    asm volatile("" : "+v"(tid_p));          //  this phase lives in registers across the main loop)
    const int lane = tid_p & 63, li = lane & 31, h = lane >> 5;
    const int gw = wc * WK + wk;
    unsigned char* myreg = smem + (gw * WP + wp) * REG;
    const int chunk = lane & 15, psub = lane >> 4;       // 16 lanes = one pixel's 64 channels, 4 pixels per instruction
    const int n_pieces = (p.proj_nks + 3) >> 2;           // (the last piece may hold 32 channels: level 2)
    auto group_sync = [&]() {
      if (NG > 1) __syncthreads();
      else cf_wave_lds_sync();
    };
    f16x8 pwh[4][RT], pwl[4][RT];
    auto load_pw = [&](int piece) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int ks = p.proj_ks0 + min(piece * 4 + kk, p.proj_nks - 1);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const unsigned char* base = p.weight + ((long)(w_ok ? rt0 + rt : 0) * p.n_ks + ks) * 2048;
          pwh[kk][rt] = *reinterpret_cast<const f16x8*>(base + (unsigned)lane * 16u);
          pwl[kk][rt] = *reinterpret_cast<const f16x8*>(base + 1024 + (unsigned)lane * 16u);
        }
      }
    };
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      for (int e0 = 0; e0 < n_pieces; e0 += NG) {
        group_sync();                        // the previous round's fragments have been read: the regions are free
        const int e = e0 + gw;
        if (e < n_pieces) {
          const int cw = min(64, p.proj_ch - e * 64);
          // (two batches of four rows: eight rows in flight beside the accumulators and the weight fragments spill)
#pragma unroll
          for (int hb = 0; hb < 2; ++hb) {
            f32x4 xv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int pl = wp * (32 * CT) + ct * 32 + (hb * 4 + j) * 4 + psub;
              int m = m0 + pl;
              bool ok = chunk * 4 < cw;
              if (T2) {
                const int y = ty0 + (pl >> 4), x = tx0 + (pl & 15);
                ok = ok && y < p.H && x < p.W;
                m = m0 + y * p.W + x;
              } else {
                ok = ok && m < p.M;
              }
              xv[j] = ok ? *reinterpret_cast<const f32x4*>(p.proj_x + (size_t)m * p.proj_c + e * 64 + chunk * 4)
                         : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const f32x4 xs = xv[j] * p.in_scale;
              uint2 hi2, lo2;
              split2(xs[0], xs[1], hi2.x, lo2.x);
              split2(xs[2], xs[3], hi2.y, lo2.y);
              unsigned char* o = myreg + ((hb * 4 + j) * 4 + psub) * BROW + chunk * 8;
              *reinterpret_cast<uint2*>(o) = hi2;
              *reinterpret_cast<uint2*>(o + BPLANE) = lo2;
            }
            asm volatile("" ::: "memory");   // (the next batch / the weight fragments are requested behind these stores)
          }
        }
        const int np = min(NG, n_pieces - e0);
        int i = wk;
        if (i < np) load_pw(e0 + i);         // (requested in front of the barrier)
        group_sync();
        for (; i < np; i += WK) {
          const unsigned char* reg = smem + (i * WP + wp) * REG + li * BROW + h * 16;
          const int nk = min(4, p.proj_nks - (e0 + i) * 4);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            if (kk < nk) {
              const f16x8 bh = *reinterpret_cast<const f16x8*>(reg + kk * 32);
              const f16x8 bl = *reinterpret_cast<const f16x8*>(reg + kk * 32 + BPLANE);
#pragma unroll
              for (int rt = 0; rt < RT; ++rt) {
                accs[rt][ct] = CF_MFMA_F16(pwl[kk][rt], bh, accs[rt][ct]);
                accm[rt][ct] = CF_MFMA_F16(pwh[kk][rt], bh, accm[rt][ct]);
                accs[rt][ct] = CF_MFMA_F16(pwh[kk][rt], bl, accs[rt][ct]);
              }
            }
          }
          if (i + WK < np) load_pw(e0 + i + WK);
        }
      }
    }
    __syncthreads();                         // every region has been read: the memory becomes the K-split / epilogue tiles
  }

  // ---- K-split waves: partial sums -> LDS, added by wave wk == 0 in fixed order
  if (WK > 1) {
    float* red = reinterpret_cast<float*>(smem);     // [wave][rt][ct][16][64]
    if (wk > 0) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            red[(((wave * RT + rt) * CT + ct) * 16 + r) * 64 + lane] = accm[rt][ct][r] + accs[rt][ct][r];
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float s = accm[rt][ct][r] + accs[rt][ct][r];
#pragma unroll
          for (int k = 1; k < WK; ++k) s += red[((((wave + k) * RT + rt) * CT + ct) * 16 + r) * 64 + lane];
          accm[rt][ct][r] = s;
          accs[rt][ct][r] = 0.0f;
        }
  }

  // ---- epilogue, coalesced form: the accumulators hold 4 consecutive channels of 32 DIFFERENT pixels per register
  // group, so storing them directly writes 32-byte pieces 64 channels apart (and reads the residual the same way) -
  // measured 8-26 % of a workgroup's time in this phase.  Each wave transposes its 32 pixels x 32*RT channels through
  // a private LDS tile instead (LDS executes a wave's instructions in order: no barrier) and then writes / reads whole
  // pixel rows: 8*RT lanes cover one contiguous run of RT*128 bytes.
  if constexpr (ROOT) {
    constexpr int EROW = RT * 128 + 16;      // transposition tile: 64 channels x 4 B + 16 per pixel row
    constexpr int BROW = 144;                // B tile: 64 f16 + 16 B per pixel row and plane
    constexpr int BPLANE = 32 * BROW;
    constexpr int REG = 2 * BPLANE;          // one region (>= 32 * EROW): 9216 B
    static_assert(REG >= 32 * EROW, "region holds a transposition tile");
    constexpr int LPP = RT * 8, PPI = 64 / LPP;
    asm volatile("; cf_epilogue_begin" ::: "memory");
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane = tid_e & 63, wave = __builtin_amdgcn_readfirstlane(tid_e >> 6), li = lane & 31, h = lane >> 5;
    const int wp = wave % WP, wc = wave / WP;   // (WK == 1)
    unsigned char* r1 = smem + wave * 2 * REG;
    unsigned char* r2 = r1 + REG;
    const int chunk = lane % LPP, psub = lane / LPP;
    const int n = wc * 64 + chunk * 4;       // this lane's 4 channels: of x2 / x1 (phases 1-2) and of the Root's output (phase 4)
    const int nl = chunk * 4;                // ... inside the wave's 64
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
    const f32x4 rbias4 = *reinterpret_cast<const f32x4*>(p.root_bias + n);
    auto group_sync = [&]() {                // the WC waves of a pixel group exchange data (WC == 1: the wave alone)
      if (WC > 1) __syncthreads();
      else cf_wave_lds_sync();
    };
    auto pixel = [&](int ct, int ploc, int& m) {     // -> inside the map?
      const int pl = wp * (32 * CT) + ct * 32 + ploc;
      if (T2) {
        const int y = ty0 + (pl >> 4), x = tx0 + (pl & 15);
        m = m0 + y * p.W + x;
        return y < p.H && x < p.W;
      }
      m = m0 + pl;
      return m < p.M;
    };
    auto put_split = [&](unsigned char* reg, int ploc, const f32x4& v) {   // 4 channels of one pixel -> B tile (hi, lo)
      const f32x4 xs = v * p.root_in_scale;
      uint2 hi2, lo2;
      split2(xs[0], xs[1], hi2.x, lo2.x);
      split2(xs[2], xs[3], hi2.y, lo2.y);
      unsigned char* o = reg + ploc * BROW + nl * 2;
      *reinterpret_cast<uint2*>(o) = hi2;
      *reinterpret_cast<uint2*>(o + BPLANE) = lo2;
    };
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      group_sync();                          // (everybody is done with the previous column tile's regions)
      // the Root's first weight fragments are requested now: they arrive under phases 1-2
      f16x8 rwh[4][RT], rwl[4][RT];          // (set ks % 4; the loop is unrolled by 4, so the set index is static)
      auto load_rw = [&](f16x8 (&dh)[RT], f16x8 (&dl)[RT], int ks) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          dh[rt] = *wfrag16(p.root_w, wc * RT + rt, ks, 0, p.root_nks, lane);
          dl[rt] = *wfrag16(p.root_w, wc * RT + rt, ks, 1, p.root_nks, lane);
        }
      };
      load_rw(rwh[0], rwl[0], 0);
      load_rw(rwh[1], rwl[1], 1);
      // 1. this convolution's accumulators -> region 1, transposed
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (accm[rt][ct][g * 4 + e] + accs[rt][ct][g * 4 + e]) * p.out_scale;
          *reinterpret_cast<f32x4*>(r1 + li * EROW + (rt * 32 + 8 * g + 4 * h) * 4) = v;
        }
      cf_wave_lds_sync();
      // 2. whole pixel rows: x2 = ReLU(conv + bias + x1) -> region 2 as a B tile; x1 stays in registers.  All rows are
      //    read (and x1 requested) before the first is written, so no LDS read follows a write inside a phase
      f32x4 x1v[32 / PPI], x2v[32 / PPI];
#pragma unroll
      for (int it = 0; it < 32 / PPI; ++it) {
        const int ploc = it * PPI + psub;
        int m;
        const bool ok = pixel(ct, ploc, m);
        x2v[it] = *reinterpret_cast<const f32x4*>(r1 + ploc * EROW + chunk * 16);
        x1v[it] = ok ? *reinterpret_cast<const f32x4*>(p.residual + (size_t)m * p.res_stride + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      cf_wave_lds_sync();                    // region 1 has been read by every lane: it becomes x1's B tile below
#pragma unroll
      for (int it = 0; it < 32 / PPI; ++it) {
        const int ploc = it * PPI + psub;
        f32x4 v = x2v[it] + bias4 + x1v[it];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
        if (p.out) {
          int m;
          if (pixel(ct, ploc, m)) *reinterpret_cast<f32x4*>(p.out + (size_t)m * p.out_stride + n) = v;
        }
        put_split(r2, ploc, v);
        put_split(r1, ploc, x1v[it]);
      }
      group_sync();
      // 3. the Root: 8 WC k-steps - x2 channels in order (the pieces of waves wc' = 0 .. WC-1 of this pixel group), then
      //    x1 channels - with the slot kernel's products in the slot kernel's order; this wave's 64 output channels
      f32x16 rm[RT], rs[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          rm[rt][r] = 0.0f;
          rs[rt][r] = 0.0f;
        }
#pragma unroll 4
      for (int ks = 0; ks < 8 * WC; ++ks) {
        const int piece = (ks >> 2) % WC, src = ks / (4 * WC);          // whose region, x2 (region 2) or x1 (region 1)
        const unsigned char* row = smem + ((piece * WP + wp) * 2 + (src == 0 ? 1 : 0)) * REG + li * BROW + (ks & 3) * 32 + h * 16;
        const f16x8 xh = *reinterpret_cast<const f16x8*>(row);
        const f16x8 xl = *reinterpret_cast<const f16x8*>(row + BPLANE);
        load_rw(rwh[(ks + 2) % 4], rwl[(ks + 2) % 4], min(ks + 2, p.root_nks - 1));   // weights two k-steps ahead
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          rs[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rwl[ks % 4][rt], xh, rs[rt], 0, 0, 0);
          rs[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rwh[ks % 4][rt], xl, rs[rt], 0, 0, 0);
          rm[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rwh[ks % 4][rt], xh, rm[rt], 0, 0, 0);
        }
      }
      // 3b. the Tree's children (further sources of the Root): 64-channel pieces, WC at a time - wave wc fetches piece
      //     e0 + wc of these 32 pixels from HBM (whole rows, as the residual), splits it into its region 2, and after the
      //     barrier every wave multiplies the round's pieces in K order.  The weight fragments keep rotating through the four sets (a
      //     piece is 4 k-steps, so the set index stays static).
      {
        const int n_extra = (p.root_nks - 8 * WC) >> 2;      // 64-channel pieces of the children
        const int ch0 = p.root_xsrc_ch[0] >> 6;               // pieces of the first child
        int ksx = 8 * WC;
        for (int e0 = 0; e0 < n_extra; e0 += WC) {
          group_sync();                        // the previous k-steps' fragments have been read: regions are free
          const int e = e0 + wc;
          if (e < n_extra) {
            // (the rows are requested here, not a round ahead or in front of the barrier: eight more row registers live
            //  across either spill; the CU's other workgroup covers the latency)
            const bool first = e < ch0;
            const float* xs = first ? p.root_xsrc[0] : p.root_xsrc[1];
            const int xc = first ? p.root_xsrc_c[0] : p.root_xsrc_c[1];
            const int off = (first ? e : e - ch0) * 64 + nl;
            f32x4 xv[32 / PPI];
#pragma unroll
            for (int it = 0; it < 32 / PPI; ++it) {
              int m;
              const bool ok = pixel(ct, it * PPI + psub, m);
              xv[it] = ok ? *reinterpret_cast<const f32x4*>(xs + (size_t)m * xc + off) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int it = 0; it < 32 / PPI; ++it) put_split(r2, it * PPI + psub, xv[it]);
          }
          group_sync();
          const int np = min(WC, n_extra - e0);
          for (int i = 0; i < np; ++i) {
            const unsigned char* reg = smem + ((i * WP + wp) * 2 + 1) * REG + li * BROW + h * 16;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              const f16x8 xh = *reinterpret_cast<const f16x8*>(reg + kk * 32);
              const f16x8 xl = *reinterpret_cast<const f16x8*>(reg + kk * 32 + BPLANE);
              load_rw(rwh[(kk + 2) % 4], rwl[(kk + 2) % 4], min(ksx + kk + 2, p.root_nks - 1));
#pragma unroll
              for (int rt = 0; rt < RT; ++rt) {
                rs[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rwl[kk][rt], xh, rs[rt], 0, 0, 0);
                rs[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rwh[kk][rt], xl, rs[rt], 0, 0, 0);
                rm[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rwh[kk][rt], xh, rm[rt], 0, 0, 0);
              }
            }
            ksx += 4;
          }
        }
      }
      group_sync();                          // every B fragment has been read: region 2 becomes the output's transposition tile
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (rm[rt][g * 4 + e] + rs[rt][g * 4 + e]) * p.root_scale;
          *reinterpret_cast<f32x4*>(r2 + li * EROW + (rt * 32 + 8 * g + 4 * h) * 4) = v;
        }
      cf_wave_lds_sync();
#pragma unroll
      for (int it = 0; it < 32 / PPI; ++it) {
        const int ploc = it * PPI + psub;
        int m;
        const bool ok = pixel(ct, ploc, m);
        f32x4 v = *reinterpret_cast<const f32x4*>(r2 + ploc * EROW + chunk * 16) + rbias4;
        if (p.root_act == CF_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
        }
        if (ok) *reinterpret_cast<f32x4*>(p.root_out + (size_t)m * p.root_out_stride + n) = v;
      }
    }
    return;
  }

  constexpr bool coalesced = WK == 1 && NT == 256;   // (8-wave configuration: measured no better; K-split waves: direct)
  if (coalesced && w_ok) {
    constexpr int EROW = RT * 128 + 16;      // bytes per pixel row of the tile: +16 B so that 16 lanes hit 64 banks
    constexpr int LPP = RT * 8;              // lanes (16-byte chunks) per pixel
    constexpr int PPI = 64 / LPP;            // pixels per instruction
    asm volatile("; cf_epilogue_begin" ::: "memory");   // marker for tools/check_isa.py (no instruction)
    // every lane-derived index of the epilogue is RE-derived here from a laundered thread id: computed once at the top of
    // the kernel they would stay live (or be spilled to scratch) across the whole MFMA loop that never uses them
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane = tid_e & 63, wave = __builtin_amdgcn_readfirstlane(tid_e >> 6), li = lane & 31, h = lane >> 5;
    const int wp = (wave / WK) % WP, wc = wave / (WK * WP);
    const int rt0 = (blockIdx.y * WC + wc) * RT;
    unsigned char* eb = smem + wave * 32 * EROW;
    const int chunk = lane % LPP, psub = lane / LPP;
    const int n = rt0 * 32 + chunk * 4;
    const bool n_ok = n < p.N;
    f32x4 bias4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (n + e < p.N) bias4[e] = p.bias[n + e];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      if (ct) cf_wave_lds_sync();            // ... and every lane has read the previous tile before it is overwritten
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (accm[rt][ct][g * 4 + e] + accs[rt][ct][g * 4 + e]) * p.out_scale;
          *reinterpret_cast<f32x4*>(eb + li * EROW + (rt * 32 + 8 * g + 4 * h) * 4) = v;
        }
      cf_wave_lds_sync();                    // the tile is complete before any lane reads another lane's part ...
#pragma unroll
      for (int it = 0; it < 32 / PPI; ++it) {
        const int ploc = it * PPI + psub;
        const int pl = wp * (32 * CT) + ct * 32 + ploc;
        int m = m0 + pl;
        bool ok = n_ok;
        if (T2) {
          const int y = ty0 + (pl >> 4), x = tx0 + (pl & 15);
          ok = ok && y < p.H && x < p.W;
          m = m0 + y * p.W + x;
        } else {
          ok = ok && m < p.M;
        }
        f32x4 v = *reinterpret_cast<const f32x4*>(eb + ploc * EROW + chunk * 16) + bias4;
        if (ok && n + 3 < p.N) {
          if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + (size_t)m * p.res_stride + n);
          if (p.act == CF_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
          }
          *reinterpret_cast<f32x4*>(p.out + (size_t)m * p.out_stride + n) = v;
        } else if (ok) {                     // last, partial group of channels (N = 27): element by element
          for (int e = 0; e < 4 && n + e < p.N; ++e) {
            float x = v[e];
            if (p.residual) x += p.residual[(size_t)m * p.res_stride + n + e];
            if (p.act == CF_ACT_RELU) x = fmaxf(x, 0.0f);
            p.out[(size_t)m * p.out_stride + n + e] = x;
          }
        }
      }
    }
  }

  // ---- epilogue, direct form (K-split waves, N not a multiple of 4): lane = pixel, register group g = 4 consecutive channels
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    if (coalesced) break;
    const int pl = wp * (32 * CT) + ct * 32 + li;
    int m = m0 + pl;
    if (T2) {
      const int y = ty0 + (pl >> 4), x = tx0 + (pl & 15);
      if (y >= p.H || x >= p.W) continue;
      m = m0 + y * p.W + x;
    } else if (m >= p.M) {
      continue;
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = (rt0 + rt) * 32 + 8 * g + 4 * h;
        if (n >= p.N) continue;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (accm[rt][ct][g * 4 + e] + accs[rt][ct][g * 4 + e]) * p.out_scale;
        if (n + 3 < p.N) {
          v += *reinterpret_cast<const f32x4*>(p.bias + n);
          if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + (size_t)m * p.res_stride + n);
          if (p.act == CF_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
          }
          *reinterpret_cast<f32x4*>(p.out + (size_t)m * p.out_stride + n) = v;
        } else {
          for (int e = 0; e < 4 && n + e < p.N; ++e) {
            float x = v[e] + p.bias[n + e];
            if (p.residual) x += p.residual[(size_t)m * p.res_stride + n + e];
            if (p.act == CF_ACT_RELU) x = fmaxf(x, 0.0f);
            p.out[(size_t)m * p.out_stride + n + e] = x;
          }
        }
      }
  }
#ifdef CF_CONV3_PROF
  PROF_MARK(3)
  if (tid == 0 && blockIdx.y == 0) {
    const int pl = 0;
    size_t m = m0 + pl;
    if (T2) m = m0 + (size_t)ty0 * p.W + tx0;
    for (int i = 0; i < 4; ++i) p.out[m * p.out_stride + i] = (float)t_prof[i];
  }
#endif
}

template <int WC, int WP, int WK, int RT, int NU, bool DB, int MINB, bool T2 = false, int CT = 2, bool S2 = false, bool ROOT = false,
          bool PROJ = false>
bool try_launch(Conv3F k, int batch, hipStream_t st) {
  constexpr int R = 32 * CT * WP, ROWB = 64 * WK + 16;
  long blocks;
  if (T2) {
    k.PR = S2 ? (2 * (R / 16) + 1) * 33 : (R / 16 + 2) * 18;
    k.tiles_x = (k.W + 15) / 16;
    k.tiles_y = (k.H + R / 16 - 1) / (R / 16);
    blocks = (long)k.tiles_x * k.tiles_y * batch;
  } else {
    k.PR = R + 2 * k.W + 2;
    blocks = (k.M + R - 1) / R;
  }
  const long units = (long)k.PR * 4 * WK;
  constexpr int NT = 64 * WC * WP * WK;
  if (units > (long)NT * NU || blocks >= (1L << 31)) return false;
  size_t dyn = (size_t)(DB ? 2 : 1) * (k.PR + 1) * ROWB;
  if (WK > 1) dyn = dyn < (size_t)4 * RT * CT * 16 * 64 * 4 ? (size_t)4 * RT * CT * 16 * 64 * 4 : dyn;
  constexpr size_t epi = (size_t)(64 * WC * WP * WK / 64) * 32 * (RT * 128 + 16);   // the waves' transposition tiles
  if (WK == 1 && NT == 256 && dyn < epi) dyn = epi;
  if (dyn > 160 * 1024) return false;
  if (S2 && MINB >= 2 && dyn > 80 * 1024) return false;
  if (ROOT) {
    constexpr size_t root_lds = (size_t)4 * 2 * 9216;   // four waves x two regions
    if (dyn < root_lds) dyn = root_lds;
    if (dyn > 80 * 1024) return false;
  }
  if (PROJ) {
    constexpr size_t proj_lds = (size_t)(NT / 64) * 9216;   // one B-tile region per wave
    if (dyn < proj_lds) dyn = proj_lds;
  }
  auto kernel = conv3x3_f16x3_kernel<WC, WP, WK, RT, NU, DB, MINB, T2, CT, S2, ROOT, PROJ>;
  static CfLdsLimit lds_limit;                // (one per template instantiation)
  lds_limit.ensure(kernel, dyn, 65536);
  const dim3 grid((unsigned)blocks, (unsigned)((k.n_rt + WC * RT - 1) / (WC * RT)));
  hipLaunchKernelGGL(kernel, grid, dim3(NT), dyn, st, k);
  return true;
}

// the default tilings of the 64+ channel layers, with or without the projection k-steps in front
template <int WC, int WP, int WK, int RT, int NU, bool DB, int MINB, bool T2 = false, int CT = 2>
bool launch_p(const Conv3F& k, int batch, hipStream_t st) {
  return k.proj_x ? try_launch<WC, WP, WK, RT, NU, DB, MINB, T2, CT, false, false, true>(k, batch, st)
                  : try_launch<WC, WP, WK, RT, NU, DB, MINB, T2, CT>(k, batch, st);
}

// 16 x 16 tiles cover the map with at most ~6 % of the tile area outside it
bool tiles_fit(int H, int W) {
  const long covered = (long)((H + 15) / 16) * 16 * ((W + 15) / 16) * 16;
  return covered * 100 <= (long)H * W * 106;
}

}  // namespace

// A geometry that fits no patch configuration is forwarded to cf_conv2d_f16x3 (same packed weights).
// root != nullptr (cf_conv3x3_root_f16x3, already validated): try the fused conv2 + Root launch first; *fused says whether it ran.
// proj_ch != nullptr (cf_conv3x3_proj_f16x3, already validated): src[1] is the 1x1 projection's source, proj_ch = the real
// channels of (the 3x3 source, the projection's source).
static int conv3x3_impl(const cf_conv_args* a, const cf_conv_args* root, const int32_t* root_ch, bool* fused, void* stream,
                        const int32_t* proj_ch = nullptr) {
  CF_REQUIRE(a != nullptr, "cf_conv3x3_f16x3: null args");
  CF_REQUIRE(a->n_src == (proj_ch ? 2 : 1) && a->src[0] && a->src_c[0] > 0 && a->src_c[0] % 4 == 0, "cf_conv3x3_f16x3: one fp32 NHWC source");
  const bool s2 = a->stride == 2;
  CF_REQUIRE((a->stride == 1 && a->Ho == a->H && a->Wo == a->W) ||
             (s2 && a->Ho == (a->H - 1) / 2 + 1 && a->Wo == (a->W - 1) / 2 + 1), "cf_conv3x3_f16x3: 3x3, pad 1, stride 1 or 2");
  if (s2 && a->residual) return cf_conv2d_f16x3(a, stream);   // (the stride-2 tile has no residual epilogue: same weights, slot kernel)
  CF_REQUIRE(a->K_pad > 0 && a->K_pad % 32 == 0, "cf_conv3x3_f16x3: K_pad=%d not a multiple of 32", a->K_pad);
  CF_REQUIRE(a->N > 0 && a->N_pad >= a->N && (a->N_pad == 32 || a->N_pad % 64 == 0), "cf_conv3x3_f16x3: N=%d N_pad=%d", a->N, a->N_pad);
  CF_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0, "cf_conv3x3_f16x3: bad geometry");
  CF_REQUIRE(a->weight && a->bias && a->out, "cf_conv3x3_f16x3: null buffer");
  CF_REQUIRE(a->out_layout == CF_LAYOUT_NHWC && a->out_stride >= a->N && a->out_stride % 4 == 0,
             "cf_conv3x3_f16x3: output must be fp32 NHWC with a stride that is a multiple of 4");
  CF_REQUIRE(a->act == CF_ACT_NONE || a->act == CF_ACT_RELU, "cf_conv3x3_f16x3: act=%d unsupported", a->act);
  CF_REQUIRE(a->out_scale > 0.0f, "cf_conv3x3_f16x3: out_scale must be the 2^-(s+4) the weights were packed with");
  CF_REQUIRE(!a->residual || a->res_stride % 4 == 0, "cf_conv3x3_f16x3: residual stride must be a multiple of 4");
  // K_pad = 16 * 9 * slices, rounded up to a multiple of 32 (+ the projection's channels behind it)
  const int slices = proj_ch ? proj_ch[0] / 16 : a->K_pad / 144;
  if (proj_ch)
    CF_REQUIRE(slices >= 2 && slices % 2 == 0 && a->K_pad == slices * 144 + proj_ch[1],
               "cf_conv3x3_proj_f16x3: K_pad=%d is not a slice-major 3x3 packing of %d channels + %d projected ones", a->K_pad, proj_ch[0], proj_ch[1]);
  else
  CF_REQUIRE(slices >= 1 && (a->K_pad == slices * 144 || a->K_pad == slices * 144 + 16),
             "cf_conv3x3_f16x3: K_pad=%d is not a slice-major 3x3 packing", a->K_pad);
  CF_REQUIRE(slices * 16 <= a->src_c[0], "cf_conv3x3_f16x3: %d input channels exceed the source width %d", slices * 16, a->src_c[0]);
  const long M = (long)a->B * a->Ho * a->Wo;
  CF_REQUIRE((long)a->B * a->H * a->W * a->src_c[0] < (1L << 31) && M < (1L << 31), "cf_conv3x3_f16x3: tensor too large");
  Conv3F k{};
  k.x = a->src[0];
  k.weight = reinterpret_cast<const unsigned char*>(a->weight);
  k.bias = a->bias; k.residual = a->residual; k.out = a->out;
  k.x_stride = a->src_c[0];
  k.H = a->Ho; k.W = a->Wo; k.HW = a->Ho * a->Wo; k.M = (int)M; k.N = a->N;
  k.Hi = a->H; k.Wi = a->W; k.HWi = a->H * a->W;
  k.n_rt = a->N_pad / 32; k.n_ks = a->K_pad / 16;
  k.res_stride = a->res_stride; k.out_stride = a->out_stride; k.act = a->act;
  k.out_scale = a->out_scale;
  k.in_scale = cf_resolve_in_scale(a->in_scale);
  CF_REQUIRE(k.in_scale > 0.0f, "cf_conv3x3_f16x3: in_scale must be 0 (= 16) or a power of two");
  if (proj_ch) {
    k.proj_x = a->src[1];
    k.proj_c = a->src_c[1];
    k.proj_ch = proj_ch[1];
    k.proj_nks = proj_ch[1] / 16;
    k.proj_ks0 = slices * 9;                 // the projection's k-steps sit behind the 3x3 part
  }
  hipStream_t st = (hipStream_t)stream;
  bool ok = false;
  const int B = a->B;
  auto cfg = [&](int WK) {
    k.n_rounds = slices / WK;
    return slices % WK == 0;
  };
  if (root && !s2 && cfg(1)) {
    Conv3F kr = k;
    kr.out = nullptr;                        // x2 stays on the chip
    kr.root_w = reinterpret_cast<const unsigned char*>(root->weight);
    kr.root_bias = root->bias;
    kr.root_out = root->out;
    kr.root_out_stride = root->out_stride;
    kr.root_act = root->act;
    kr.root_scale = root->out_scale;
    kr.root_in_scale = cf_resolve_in_scale(root->in_scale);
    CF_REQUIRE(kr.root_in_scale > 0.0f, "cf_conv3x3_root_f16x3: the Root's in_scale must be 0 (= 16) or a power of two");
    kr.root_nks = root->K_pad / 16;
    for (int i = 0; i < 2; ++i) {
      kr.root_xsrc[i] = i + 2 < root->n_src ? root->src[i + 2] : nullptr;
      kr.root_xsrc_c[i] = i + 2 < root->n_src ? root->src_c[i + 2] : 0;
      kr.root_xsrc_ch[i] = i + 2 < root->n_src ? root_ch[i + 2] : 0;
    }
    const bool t2r = (long)a->H * a->W >= 4096 && tiles_fit(a->H, a->W);
    const long half_tiles = (long)((a->H + 7) / 8) * ((a->W + 15) / 16) * B;
    bool okr = false;
    // small launches (latency chains of bs = 1 ... 4): the half-height tiles of the unfused path, fused - one launch less
    // where the whole chain is a handful of under-filled launches
    if (a->N_pad == 64 && half_tiles <= 256)
      okr = try_launch<1, 4, 1, 2, 4, true, 2, true, 1, false, true>(kr, B, st);
    else if (a->N_pad == 256 && (long)a->H * a->W > 512 && (M + 31) / 32 <= 256)
      okr = try_launch<4, 1, 1, 2, 4, true, 2, false, 1, false, true>(kr, B, st);
    else if (a->N_pad == 64 && half_tiles > 256)
      okr = (t2r && try_launch<1, 4, 1, 2, 6, true, 2, true, 2, false, true>(kr, B, st)) ||
            try_launch<1, 4, 1, 2, 6, true, 2, false, 2, false, true>(kr, B, st);
    else if (a->N_pad == 128 && 2 * half_tiles > 256)
      okr = (t2r && try_launch<2, 2, 1, 2, 4, true, 2, true, 2, false, true>(kr, B, st)) ||
            try_launch<2, 2, 1, 2, 6, true, 2, false, 2, false, true>(kr, B, st) ||
            try_launch<2, 2, 1, 2, 4, true, 2, true, 2, false, true>(kr, B, st);
    else if (a->N_pad == 256 && (long)a->H * a->W > 512 && (M + 31) / 32 > 256)
      okr = try_launch<4, 1, 1, 2, 4, true, 2, false, 2, false, true>(kr, B, st) ||
            try_launch<4, 1, 1, 2, 2, true, 2, true, 2, false, true>(kr, B, st);
    if (okr) {
      *fused = true;
      return cf_check_launch("cf_conv3x3_root_f16x3");
    }
  }
  if (s2) {
    // stride 2: 4 x 16 or 8 x 16 output tiles by output width; a geometry the patch does not fit goes to the slot kernel
    k.n_rounds = slices;
    static const int alt = [] { const char* e = getenv("CF_CONV3_S2_ALT"); return e ? atoi(e) : 0; }();   // (dev A/B: other tile heights)
    if (a->N_pad == 64) ok = try_launch<1, 4, 1, 2, 9, false, 2, true, 1, true>(k, a->B, st);
    else if (a->N_pad == 128) ok = (alt & 1) ? try_launch<2, 2, 1, 2, 9, false, 2, true, 2, true>(k, a->B, st)
                                             : try_launch<2, 2, 1, 2, 5, true, 2, true, 1, true>(k, a->B, st);
    else if (a->N_pad >= 256) {
      // half-height tiles (2 x 16) while that grid still fits the chip in one round (level 5 of a bs <= 8 launch: 34.5 vs
      // 47.9 us); same sums whatever the tile, so this may depend on the batch size (as the stride-1 small-grid rule)
      const long half_tiles = (long)((a->Ho + 1) / 2) * ((a->Wo + 15) / 16) * a->B * ((a->N_pad + 255) / 256);
      ok = ((alt & 2) || half_tiles <= 256) ? try_launch<4, 1, 1, 2, 3, true, 2, true, 1, true>(k, a->B, st)
           : (alt & 4)                      ? try_launch<4, 2, 1, 2, 5, true, 1, true, 1, true>(k, a->B, st)
                                            : try_launch<4, 1, 1, 2, 5, true, 2, true, 2, true>(k, a->B, st);
    }
    if (!ok) return cf_conv2d_f16x3(a, stream);
    return cf_check_launch("cf_conv3x3_f16x3");
  }
  // dev override (tools/bench_conv_cfg.py): CF_CONV3_CFG="WC,WP,WK[,T2]" forces one of the instantiated tilings
  static const int only_n = [] { const char* e = getenv("CF_CONV3_ONLY_N"); return e ? atoi(e) : 0; }();   // (dev: override one width only)
  if (const char* force = (!proj_ch && (only_n == 0 || only_n == a->N_pad)) ? getenv("CF_CONV3_CFG") : nullptr) {
    int wc = 0, wp = 0, wk = 0, t2f = 0, ct = 2, rt = 2;
    if (sscanf(force, "%d,%d,%d,%d,%d,%d", &wc, &wp, &wk, &t2f, &ct, &rt) >= 3 && cfg(wk)) {
      const int key = wc * 100 + wp * 10 + wk;
      bool done = false;
      if (a->N_pad >= 64 && ct == 1) {              // half-size pixel tiles (32 pixels per wave): small grids, see below
        switch (key) {
          case 221: done = try_launch<2, 2, 1, 2, 4, true, 2, false, 1>(k, B, st); break;
          case 411: done = try_launch<4, 1, 1, 2, 4, true, 2, false, 1>(k, B, st); break;
          case 212: done = try_launch<2, 1, 2, 2, 4, true, 2, false, 1>(k, B, st); break;
          case 141: done = t2f ? try_launch<1, 4, 1, 2, 4, true, 2, true, 1>(k, B, st) : try_launch<1, 4, 1, 2, 6, true, 2, false, 1>(k, B, st); break;
          default: break;
        }
      } else
#ifdef CF_ONESET
      if (a->N_pad >= 64 && ct == 4 && rt == 2) {   // 64-channel x 128-pixel wave tiles at TWO waves per SIMD (one accumulator set)
        switch (key) {
          case 141: done = t2f && try_launch<1, 4, 1, 2, 10, true, 2, true, 4>(k, B, st); break;
          case 221: done = t2f ? try_launch<2, 2, 1, 2, 6, true, 2, true, 4>(k, B, st) : try_launch<2, 2, 1, 2, 8, true, 2, false, 4>(k, B, st); break;
          case 411: done = t2f ? try_launch<4, 1, 1, 2, 4, true, 2, true, 4>(k, B, st) : try_launch<4, 1, 1, 2, 4, true, 2, false, 4>(k, B, st); break;
          default: break;
        }
      } else
#endif
      if (a->N_pad >= 64 && ct == 4 && rt == 1) {   // 32-channel x 128-pixel wave tiles, two waves per SIMD: half the weight stream
        switch (key) {
          case 221: done = t2f && try_launch<2, 2, 1, 1, 6, true, 2, true, 4>(k, B, st); break;   // (the flat forms of these two spill)
          case 411: done = t2f && try_launch<4, 1, 1, 1, 4, true, 2, true, 4>(k, B, st); break;
          case 421: done = t2f ? try_launch<4, 2, 1, 1, 4, true, 1, true, 4>(k, B, st) : try_launch<4, 2, 1, 1, 4, true, 1, false, 4>(k, B, st); break;
          default: break;
        }
      } else if (a->N_pad >= 64 && ct == 4) {       // one wave per SIMD, 64 x 128 wave tiles
        switch (key) {
          case 141: done = t2f ? try_launch<1, 4, 1, 2, 10, true, 1, true, 4>(k, B, st) : try_launch<1, 4, 1, 2, 16, true, 1, false, 4>(k, B, st); break;
          case 221: done = try_launch<2, 2, 1, 2, 8, true, 1, false, 4>(k, B, st); break;
          case 411: done = try_launch<4, 1, 1, 2, 4, true, 1, false, 4>(k, B, st); break;
          default: break;
        }
      } else if (a->N_pad >= 64) {
        switch (key) {
          case 221: done = try_launch<2, 2, 1, 2, 6, true, 2>(k, B, st); break;
          case 212: done = try_launch<2, 1, 2, 2, 4, true, 2>(k, B, st); break;
          case 411: done = try_launch<4, 1, 1, 2, 4, true, 2>(k, B, st); break;
          case 421: done = try_launch<4, 2, 1, 2, 2, true, 1>(k, B, st); break;
          case 141: done = t2f ? try_launch<1, 4, 1, 2, 6, true, 2, true>(k, B, st) : try_launch<1, 4, 1, 2, 6, true, 2>(k, B, st); break;
          default: break;
        }
      }
      if (done) return cf_check_launch("cf_conv3x3_f16x3");
    }
  }
  const bool big = (long)a->H * a->W >= 4096;     // (never a function of the batch size: WK changes the
                                                  //  summation order, and a shard of a batch has to
                                                  //  reproduce the full batch bit for bit)
  const bool t2 = big && tiles_fit(a->H, a->W);
  // SMALL GRIDS (small batches): while a launch of half-size pixel tiles (CT = 1: 32 pixels per wave) still fits the chip
  // in one round - at most one workgroup per CU - it is a third faster than the default tile, whose handful of
  // workgroups each walk twice the MFMAs per round on an otherwise empty chip (bs=1, 128 -> 128 at 56x100: 16.6 vs
  // 25.8 us; 256 -> 256 at 28x50: 27.5 vs 41.2 us; 512 -> 512 at 14x25: 31.8 vs 43.2 us; at 262 workgroups the gain is
  // gone).  The tile shape changes no sum (same K order, same WK): results are bit-identical, so this MAY depend on the
  // batch size; the patch is tiled (8 x 16) whatever the map - tile quantisation costs nothing on an empty chip.
  const long tiles8x16 = (long)((a->H + 7) / 8) * ((a->W + 15) / 16) * B;
  static const long round_wgs = [] { const char* e = getenv("CF_CONV3_ONE_ROUND"); return e ? atol(e) : 256L; }();   // (dev A/B of the threshold)
  auto one_round = [&](long wgs) { return wgs <= round_wgs; };
  if (a->N_pad == 32) {
    if (big && cfg(1) && one_round(tiles8x16)) ok = try_launch<1, 4, 1, 1, 4, true, 2, true, 1>(k, B, st);
    if (ok) {
    } else if (big && cfg(1)) {
      ok = (t2 && try_launch<1, 4, 1, 1, 6, true, 2, true>(k, B, st)) || try_launch<1, 4, 1, 1, 8, true, 2>(k, B, st) ||
           try_launch<1, 4, 1, 1, 12, true, 1>(k, B, st);
    } else {
      if (cfg(4)) ok = try_launch<1, 1, 4, 1, 8, true, 2>(k, B, st) || try_launch<1, 1, 4, 1, 12, true, 2>(k, B, st);
      if (!ok && cfg(2)) ok = try_launch<1, 2, 2, 1, 12, true, 2>(k, B, st);
      if (!ok && cfg(1)) ok = try_launch<1, 4, 1, 1, 12, true, 1>(k, B, st);
    }
  } else if (a->N_pad == 64) {
    if (cfg(1)) ok = (one_round(tiles8x16) && launch_p<1, 4, 1, 2, 4, true, 2, true, 1>(k, B, st)) ||
                     (t2 && launch_p<1, 4, 1, 2, 6, true, 2, true>(k, B, st)) || launch_p<1, 4, 1, 2, 6, true, 2>(k, B, st);
  } else if (a->N_pad == 128) {
    // (the one-wave-per-SIMD form - CT = 4, 64-channel x 128-pixel wave tiles, accumulators in AGPRs, CF_CONV3_CFG
    //  "2,2,1,0,4" - is bit-identical and measured 97-123 us against 76-92 us here: DESIGN.md section 9)
    // wide maps (3x896x1600: level 3 is 112 x 200): 8 x 16 tiles with a frame - the flat run's patch (R + 2W + 2 rows)
    // no longer fits, and falling back to the slot kernel re-gathers every input 9 times
    if (cfg(1)) ok = (one_round(2 * tiles8x16) && launch_p<1, 4, 1, 2, 4, true, 2, true, 1>(k, B, st)) ||   // (64 channels per workgroup)
                     (t2 && launch_p<2, 2, 1, 2, 4, true, 2, true>(k, B, st)) || launch_p<2, 2, 1, 2, 6, true, 2>(k, B, st) ||
                     launch_p<2, 2, 1, 2, 4, true, 2, true>(k, B, st);   // (8 waves x 256 pixels measured 4 % slower here)
  } else {
    // 256+ channels: every 64-pixel tile streams the whole weight matrix from L2, which bounds these
    // layers - with enough tiles to go round, 8 waves (two pixel groups per channel group, 128 pixels)
    // halve that stream (the second group's fragment loads hit L1)
    const long tiles128 = (M + 127) / 128 * ((a->N_pad + 255) / 256);
    // smallest maps (level5, 14x25): 128 channels x 64 pixels per workgroup with K split over wave pairs
    // doubles the workgroup count and halves every wave's round chain (per-image rule, see above)
    const long runs32 = (M + 31) / 32;
    if ((long)a->H * a->W <= 512 && cfg(2))
      ok = (one_round(runs32 * ((a->N_pad + 127) / 128)) && launch_p<2, 1, 2, 2, 4, true, 2, false, 1>(k, B, st)) ||
           launch_p<2, 1, 2, 2, 4, true, 2>(k, B, st);
    if (!ok && cfg(1) && one_round(runs32 * ((a->N_pad + 255) / 256))) ok = launch_p<4, 1, 1, 2, 4, true, 2, false, 1>(k, B, st);
    // (the tiled forms behind the flat ones: maps wider than ~60 / ~95 pixels, e.g. level 4 of a 3x896x1600 input)
    if (!ok && cfg(1)) ok = (tiles128 >= 160 && (launch_p<4, 2, 1, 2, 2, true, 1>(k, B, st) || launch_p<4, 2, 1, 2, 2, true, 1, true>(k, B, st))) ||
                            launch_p<4, 1, 1, 2, 4, true, 2>(k, B, st) || launch_p<4, 1, 1, 2, 2, true, 2, true>(k, B, st);
  }
  if (!ok) return cf_conv2d_f16x3(a, stream);
  return cf_check_launch("cf_conv3x3_f16x3");
}

extern "C" int cf_conv3x3_f16x3(const cf_conv_args* a, void* stream) { return conv3x3_impl(a, nullptr, nullptr, nullptr, stream); }

// BasicBlock conv2 whose residual is the Tree's `project` (1x1 convolution + BN of the pooled level input, dla.py:96-107):
// src[1] = the pooled tensor, its k-steps packed behind the 3x3 part (packing.pack_conv_f16(proj=...)); one launch, the
// residual tensor never exists.  Geometries no patch tiling fits run the slot kernel on the same table.
extern "C" int cf_conv3x3_proj_f16x3(const cf_conv_args* a, const int32_t* src_channels, void* stream) {
  CF_REQUIRE(a != nullptr && src_channels != nullptr, "cf_conv3x3_proj_f16x3: null args");
  CF_REQUIRE(a->n_src == 2 && a->src[0] && a->src[1], "cf_conv3x3_proj_f16x3: two sources (3x3 input, projected tensor)");
  CF_REQUIRE(a->stride == 1 && a->N_pad >= 64, "cf_conv3x3_proj_f16x3: stride 1, 64+ output channels");
  CF_REQUIRE(src_channels[0] >= 32 && src_channels[0] % 32 == 0 && src_channels[0] <= a->src_c[0] && a->src_c[0] % 8 == 0,
             "cf_conv3x3_proj_f16x3: %d channels of the 3x3 source (width %d)", src_channels[0], a->src_c[0]);
  CF_REQUIRE(src_channels[1] >= 32 && src_channels[1] % 32 == 0 && src_channels[1] <= a->src_c[1] && a->src_c[1] % 8 == 0,
             "cf_conv3x3_proj_f16x3: %d channels of the projected source (width %d)", src_channels[1], a->src_c[1]);
  CF_REQUIRE((long)a->B * a->H * a->W * a->src_c[1] < (1L << 31), "cf_conv3x3_proj_f16x3: tensor too large");
  CF_REQUIRE(a->slots != nullptr, "cf_conv3x3_proj_f16x3: the slot table is needed (fallback)");
  return conv3x3_impl(a, nullptr, nullptr, nullptr, stream, src_channels);
}

// BasicBlock conv2 (+ residual + ReLU) and the Tree's Root over (conv2's output, conv2's residual) as ONE launch where
// a workgroup holds every channel of its pixels (64-channel layers); everything else runs the two launches.  Same
// bits either way (the Root's products and their order are the slot kernel's).
extern "C" int cf_conv3x3_root_f16x3(const cf_conv_args* a, const cf_conv_args* r, const int32_t* root_channels, void* stream) {
  CF_REQUIRE(a != nullptr && r != nullptr && root_channels != nullptr, "cf_conv3x3_root_f16x3: null args");
  CF_REQUIRE(a->out && a->residual && a->act == CF_ACT_RELU && a->stride == 1,
             "cf_conv3x3_root_f16x3: conv2 needs its output buffer (fallback), a residual and ReLU");
  CF_REQUIRE(r->n_src >= 2 && r->n_src <= CF_MAX_SRC && r->src[0] == a->out && r->src[1] == a->residual &&
                 r->src_c[0] == a->out_stride && r->src_c[1] == a->res_stride,
             "cf_conv3x3_root_f16x3: the Root's first sources must be (conv2's output, conv2's residual)");
  CF_REQUIRE(r->B == a->B && r->H == a->Ho && r->W == a->Wo && r->Ho == r->H && r->Wo == r->W && r->stride == 1,
             "cf_conv3x3_root_f16x3: the Root is a 1x1 convolution on conv2's output map");
  CF_REQUIRE(r->weight && r->bias && r->out && !r->residual && r->out_scale > 0.0f &&
                 (r->act == CF_ACT_NONE || r->act == CF_ACT_RELU) && r->out_layout == CF_LAYOUT_NHWC &&
                 r->out_stride >= r->N && r->out_stride % 4 == 0,
             "cf_conv3x3_root_f16x3: bad Root argument block");
  int k_sum = 0;
  bool pieces_ok = root_channels[0] == a->N && root_channels[1] == a->N;
  for (int i = 0; i < r->n_src; ++i) {
    CF_REQUIRE(r->src[i] && root_channels[i] > 0 && root_channels[i] <= r->src_c[i], "cf_conv3x3_root_f16x3: source %d invalid", i);
    k_sum += root_channels[i];
    if (i >= 2) pieces_ok = pieces_ok && root_channels[i] % 64 == 0 && r->src_c[i] % 4 == 0;
  }
  CF_REQUIRE(k_sum <= r->K_pad, "cf_conv3x3_root_f16x3: the sources' channels (%d) exceed K_pad = %d", k_sum, r->K_pad);
  bool fused = false;
  const bool fusable = (a->N == 64 || a->N == 128 || a->N == 256) && a->N_pad == a->N && r->N == a->N &&
                       r->N_pad == a->N && pieces_ok && r->K_pad == k_sum && a->res_stride % 4 == 0 && r->out_stride % 4 == 0;
  static const int fuse_on = [] { const char* e = getenv("CF_ROOT_FUSE"); return e ? atoi(e) : 1; }();   // (dev A/B: 0 off, 1 on, 2 on without children)
  const bool want = fusable && fuse_on && (fuse_on == 1 || r->n_src == 2);
  const int rc = conv3x3_impl(a, want ? r : nullptr, root_channels, &fused, stream);
  if (rc != CF_OK || fused) return rc;
  return cf_conv2d_f16x3(r, stream);
}
