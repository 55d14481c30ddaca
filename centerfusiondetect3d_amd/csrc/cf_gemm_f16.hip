// Implicit-GEMM convolution with fp32 STORAGE and split-fp16 COMPUTE ("f16x3") on the gfx950 f16 MFMA
// pipe - the fast path for the backbone / offset convolutions, whose rounding must stay at fp32 level
// because the DCN neck amplifies it ~100x (DESIGN.md §4 Numerics).
//
// Numerics.  An fp32 operand x (times a power-of-two scale that keeps its low part out of the fp16
// subnormal range) is split into hi = rne_f16(x), lo = rne_f16(x - hi): hi + lo carries ~22-24
// significant bits.  A product is  a_hi*b_hi + (a_lo*b_hi + a_hi*b_lo)  - three
// v_mfma_f32_32x32x16_f16 per 16-deep k-step (the dropped a_lo*b_lo term is < 2^-22 relative).  Each
// MFMA sums its 16 products before ONE fp32 rounding into the accumulator, so the rounding chain is
// K/16 long instead of K (fp32 MFMA 32x32x2: K/2); the two small cross terms go to their own
// accumulator so they do not add rounding steps to the main sum.  CPU emulation and the measured
// end-to-end error put this at the level of the two-level fp32 path (tools/stage_error.py).
// Scales: weights are pre-multiplied by 2^s per layer (host, max|w| -> ~2^14), activations by 2^4 at
// staging; the epilogue multiplies by 2^-(s+4) - all exact.  Activations must satisfy |x| < 4094
// (fp16 range after the 2^4 scale); they are clamped there, which only matters for absurd inputs.
//
// Structure (same SWAPPED orientation as cf_heads.hip): MFMA A-operand = weights, pre-packed on the
// host in fragment order and read straight from L2 (2 k-steps ahead in registers, no LDS, no
// barrier); B-operand = pixels: the fp32 NHWC activations are gathered per 8-channel slot, split to
// fp16 hi/lo on the fly and staged through a double-buffered LDS tile [64*WP px][32 k] - one barrier
// per 32-deep chunk.  Workgroup = 4 waves as WC (channel groups) x WP (pixel groups); a wave owns
// RT*32 output channels x 64 pixels.  Accumulators have pixels on lanes and 4 consecutive channels
// per register group, so the fp32 NHWC store is one 16-byte write per lane and group.
#include <stdlib.h>
#include "cf_f16x3.h"
#include "cf_mx.h"

namespace {

struct ConvF {
  const float* src[CF_MAX_SRC];
  int src_c[CF_MAX_SRC];
  const unsigned char* weight;  // [N_pad/32][K_pad/16][2][64][8 f16]
  const cf_slot* slots;         // 8-channel slots, 4 per chunk
  const float* bias;
  const float* residual;
  float* out;
  int H, W, Ho, Wo, stride, n_chunks, res_stride, out_stride, act, M, N, HoWo, n_rt;
  float out_scale;
  float in_scale;               // activation pre-scale (power of two)
};

// DB: double-buffered pixel tile (one barrier per chunk).  The 256-pixel tile of the 64-channel
// layers (WP = 4) is single-buffered (two barriers per chunk) so two workgroups still fit a CU.
template <int WC, int WP, int RT, bool DB>
__global__ __launch_bounds__(256, 2) void conv_f16x3_kernel(ConvF p) {
  static_assert(WC * WP == 4, "4 waves per workgroup");
  constexpr int PXB = 64 * WP;             // pixels per workgroup
  constexpr int PLANE = PXB * FROWB;       // bytes per plane of one chunk buffer
  constexpr int BUF = 2 * PLANE;
  constexpr int EROW = RT * 128 + 16;      // epilogue: a wave's 32 pixels x 32*RT channels, transposed through LDS
  constexpr int SMEM = (DB ? 2 : 1) * BUF > 4 * 32 * EROW ? (DB ? 2 : 1) * BUF : 4 * 32 * EROW;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  extern __shared__ __attribute__((aligned(16))) cf_slot lds_slots[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: scalar tile / weight addressing)
  const int li = lane & 31, h = lane >> 5;
  const int wc = wave / WP, wp = wave % WP;
  const int m0 = cf_xcd_remap(blockIdx.x, gridDim.x) * PXB;   // consecutive pixel tiles share an XCD (L2)
  const int rt0 = (blockIdx.y * WC + wc) * RT;          // first 32-row tile of this wave
  const bool w_ok = rt0 < p.n_rt;                       // (RT divides the padded tile count)
  const int n_ks = p.n_chunks * 2;
  for (int i = tid; i < p.n_chunks * 4; i += 256) lds_slots[i] = p.slots[i];

  // staging role: WP (pixel, 8-channel unit) pairs per thread
  int y0[WP], x0[WP], boff[WP];
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    const int px = (tid + 256 * i) >> 2;
    const int m = m0 + px;
    if (m < p.M) {
      const int b = m / p.HoWo, rem = m - b * p.HoWo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      y0[i] = ho * p.stride;
      x0[i] = wo * p.stride;
      boff[i] = b * p.H * p.W;
    } else {
      y0[i] = -(1 << 28);
      x0[i] = 0;
      boff[i] = 0;
    }
  }
  __syncthreads();

  f32x4 raw[WP][2];
  auto load_b = [&](int c) {
    const int src = __builtin_amdgcn_readfirstlane(lds_slots[c * 4].src);
    const float* sp = src == 1 ? p.src[1] : src == 2 ? p.src[2] : src == 3 ? p.src[3] : p.src[0];
    const int sc = src == 1 ? p.src_c[1] : src == 2 ? p.src_c[2] : src == 3 ? p.src_c[3] : p.src_c[0];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      const cf_slot s = lds_slots[c * 4 + (tid & 3)];
      const int y = y0[i] + s.dy, x = x0[i] + s.dx;
      const bool ok = (s.c_off >= 0) && ((unsigned)y < (unsigned)p.H) && ((unsigned)x < (unsigned)p.W);
      raw[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      raw[i][1] = raw[i][0];
      if (ok) {
        const float* a = sp + (size_t)(boff[i] + y * p.W + x) * sc + s.c_off;
        raw[i][0] = *reinterpret_cast<const f32x4*>(a);
        raw[i][1] = *reinterpret_cast<const f32x4*>(a + 4);
      }
    }
  };
  auto store_b = [&](unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      const int pr = tid + 256 * i;
      u32x4 hi, lo;
      split8(raw[i][0], raw[i][1], hi, lo, p.in_scale);
      unsigned char* o = buf + (pr >> 2) * FROWB + (pr & 3) * 16;
      *reinterpret_cast<u32x4*>(o) = hi;
      *reinterpret_cast<u32x4*>(o + PLANE) = lo;
    }
  };

  f32x16 accm[RT][2], accs[RT][2];   // main (hi*hi) and small (lo*hi + hi*lo) sums
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        accm[a][b][r] = 0.0f;
        accs[a][b][r] = 0.0f;
      }

  f16x8 wh[2][RT], wl[2][RT];        // weight fragments of k-steps (2n) and (2n+1)
  auto load_w = [&](f16x8 (&dh)[RT], f16x8 (&dl)[RT], int ks) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      dh[rt] = *wfrag16(p.weight, w_ok ? rt0 + rt : 0, ks, 0, n_ks, lane);
      dl[rt] = *wfrag16(p.weight, w_ok ? rt0 + rt : 0, ks, 1, n_ks, lane);
    }
  };
  auto mma_kstep = [&](const unsigned char* buf, int s, const f16x8 (&ah)[RT], const f16x8 (&al)[RT]) {
    f16x8 xh[2], xl[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const unsigned char* row = buf + (wp * 64 + ct * 32 + li) * FROWB + s * 32 + h * 16;
      xh[ct] = *reinterpret_cast<const f16x8*>(row);
      xl[ct] = *reinterpret_cast<const f16x8*>(row + PLANE);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        accs[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[rt], xh[ct], accs[rt][ct], 0, 0, 0);
        accs[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[rt], xl[ct], accs[rt][ct], 0, 0, 0);
        accm[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[rt], xh[ct], accm[rt][ct], 0, 0, 0);
      }
  };

  load_b(0);
  load_w(wh[0], wl[0], 0);
  load_w(wh[1], wl[1], 1);
  store_b(smem);
  load_b(p.n_chunks > 1 ? 1 : 0);
  __syncthreads();
  // Straight-line body (indices clamped instead of branches) so the scheduler may interleave the
  // VALU operand split of chunk c+1 with the MFMAs of chunk c (sched_group_barrier pattern below).
  const int last = p.n_chunks - 1;
  for (int c = 0; c < p.n_chunks; ++c) {
    unsigned char* cur = smem + (DB ? (c & 1) * BUF : 0);
    unsigned char* nxt = smem + (DB ? ((c + 1) & 1) * BUF : 0);
    mma_kstep(cur, 0, wh[0], wl[0]);
    load_w(wh[0], wl[0], min(2 * c + 2, n_ks - 2));
    if (!DB) {
      mma_kstep(cur, 1, wh[1], wl[1]);
      load_w(wh[1], wl[1], min(2 * c + 3, n_ks - 1));
      __syncthreads();                      // single buffer: everyone is done reading it
      store_b(nxt);
    } else {
      store_b(nxt);                         // chunk c+1 (requested one chunk ago): VALU split + 2 LDS writes
      mma_kstep(cur, 1, wh[1], wl[1]);
      load_w(wh[1], wl[1], min(2 * c + 3, n_ks - 1));
#pragma unroll
      for (int i = 0; i < 6 * RT; ++i) {    // pair every MFMA of the chunk with a few VALU / DS ops
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 4 * WP, 0);   // VALU
      }
    }
    load_b(min(c + 2, last));
    __syncthreads();
  }

  // ---- epilogue, coalesced (see cf_conv3x3_f16.hip): each wave transposes its 32 pixels x 32*RT channels through a
  // private LDS tile (free after the loop's last barrier) and stores / reads the residual as whole pixel rows
  if (w_ok) {
    constexpr int LPP = RT * 8, PPI = 64 / LPP;
    asm volatile("; cf_epilogue_begin" ::: "memory");   // marker for tools/check_isa.py (no instruction)
    unsigned char* eb = smem + wave * 32 * EROW;
    const int chunk = lane % LPP, psub = lane / LPP;
    const int n = rt0 * 32 + chunk * 4;
    f32x4 bias4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (n + e < p.N) bias4[e] = p.bias[n + e];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
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
        const size_t m = (size_t)m0 + wp * 64 + ct * 32 + ploc;
        f32x4 v = *reinterpret_cast<const f32x4*>(eb + ploc * EROW + chunk * 16) + bias4;
        if (m >= (size_t)p.M || n >= p.N) continue;
        if (n + 3 < p.N) {
          if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + m * p.res_stride + n);
          if (p.act == CF_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
          }
          *reinterpret_cast<f32x4*>(p.out + m * p.out_stride + n) = v;
        } else {                             // last, partial group of channels: element by element
          for (int e = 0; e < 4 && n + e < p.N; ++e) {
            float x = v[e];
            if (p.residual) x += p.residual[m * p.res_stride + n + e];
            if (p.act == CF_ACT_RELU) x = fmaxf(x, 0.0f);
            p.out[m * p.out_stride + n + e] = x;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// DCNv2 main GEMM on the same f16x3 scheme.  The B operand is the bilinear sample: per (pixel, tap)
// the sampling position and 16*sigmoid(mask) are computed once per tile into LDS; per chunk a thread
// requests the 4 corner rows (8 fp32 channels each) of its (pixel, unit) pairs one chunk ahead,
// combines them in fp32, splits to fp16 hi/lo and stages them.  The VALU work per sample (~14
// instructions) is what bounds the 64-output-channel layers, not the MFMA pipe.
// ---------------------------------------------------------------------------------------------
struct DcnF {
  const float* x;
  const float* om;
  const unsigned char* weight;
  const float* bias;
  float* out;
  int om_stride, H, W, C, n_chunks, chunks_per_tap, out_stride, act, M, N, n_rt;
  float out_scale;
  float in_scale;        // activation pre-scale (power of two), carried by the modulation factor
  float mx_scale;        // pre-scale of the mx rows (out_mx)
  unsigned* out_split;   // optional split-bf16 copy [M][2][split_stride] (as 32-bit words: 2 bf16 each)
  int split_stride;
  unsigned char* out_mx; // optional mx rows [M][272] (cf_pack_feat_mx's format; N = 64, one 32-channel row tile per wave)
  float* partial;        // K split (gridDim.z > 1): raw partial sums [z][M][n_rt * 32], reduced by dcn_reduce_kernel
  int direct_epilogue;   // dev A/B (CF_DCN_EPI=0): store the accumulators directly
  int mask_activated;    // offmask channels 18..26 are modulation factors already (no sigmoid here)
};

// CT = 32-pixel column tiles per wave: 2 (64 pixels per wave), or 1 - half-size pixel tiles: half the accumulators and half the corner
// registers per thread, half the LDS: three workgroups per CU instead of two where a wave's chain has nothing to hide behind.
template <int WC, int WP, int RT, bool COAL, int CT = 2>
__global__ __launch_bounds__(256, 2) void dcn_f16x3_kernel(DcnF p) {
  static_assert(WC * WP == 4, "4 waves per workgroup");
  static_assert(CT == 2 || (CT == 1 && WP == 2), "half-size tiles: two pixel groups of 32");
  constexpr int PXB = 32 * CT * WP;
  constexpr int NP = PXB * 4 / 256;          // (pixel, 8-channel unit) pairs per thread and chunk
  constexpr int PLANE = PXB * FROWB;
  constexpr int BUF = 2 * PLANE;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF + PXB * 9 * 32];
  // sampling descriptor of one (pixel, tap), built ONCE per tile:
  //   dA = {element offset of the top-left corner (clamped into the image), step to the right corner
  //         (0 or C), step to the bottom corner (0 or W*C), 16 * sigmoid(mask)}
  //   dB = the four bilinear weights, ZERO where the corner lies outside the image
  // so the per-chunk staging is 8 unconditional loads (every address valid), 4 multiply-adds per
  // channel and the split - no floor / compare / branch in the K loop.
  f32x4* desc = reinterpret_cast<f32x4*>(smem + 2 * BUF);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: scalar tile / weight addressing)
  const int li = lane & 31, h = lane >> 5;
  const int wc = wave / WP, wp = wave % WP;
  // consecutive pixel tiles on ONE XCD: the gathered rows of a tile and of its neighbours then meet
  // in that XCD's L2 instead of being fetched by all eight (hardware deals workgroups round-robin)
  const int m0 = cf_xcd_remap(blockIdx.x, gridDim.x) * PXB;
  const int rt0 = (blockIdx.y * WC + wc) * RT;
  const bool w_ok = rt0 < p.n_rt;
  const int n_ks = p.n_chunks * 2;
  const int HW = p.H * p.W;
#ifdef CF_DCN_PROF   // dev (tools/prof_dcn.py): cycles per phase of thread 0, written over its first output values
  long long t_prof[4] = {0, 0, 0, 0};
  long long t_last = clock64();
#define DPROF_MARK(i) { const long long t_now = clock64(); t_prof[i] += t_now - t_last; t_last = t_now; }
#else
#define DPROF_MARK(i)
#endif

  // (all offset / mask values of the tile are requested before the first is used: one memory round trip for the phase
  //  instead of one per descriptor - it was 13 % of a 64-channel layer's workgroup time)
  constexpr int NDI = (PXB * 9 + 255) / 256;
  float omy[NDI], omx[NDI], omm[NDI];
#ifdef CF_DCN_NODESC        // (dev timing experiment: no descriptor phase - every sample is the pixel's own cell with weight 1, 0, 0, 0)
  for (int i = tid; i < PXB * 9; i += 256) {
    const int m = min(m0 + i / 9, p.M - 1);
    desc[2 * i] = f32x4{__int_as_float(m * p.C), __int_as_float(0), __int_as_float(0), p.in_scale};
    desc[2 * i + 1] = f32x4{1.0f, 0.0f, 0.0f, 0.0f};
  }
  if (false)
#endif
#pragma unroll
  for (int it = 0; it < NDI; ++it) {
    const int i = min(tid + 256 * it, PXB * 9 - 1);
    const int r = i / 9, tap = i - r * 9;
    const float* om = p.om + (size_t)min(m0 + r, p.M - 1) * p.om_stride;
    omy[it] = om[2 * tap];
    omx[it] = om[2 * tap + 1];
    omm[it] = om[18 + tap];
  }
#ifdef CF_DCN_NODESC
  if (false)
#endif
#pragma unroll
  for (int it = 0; it < NDI; ++it) {
    const int i = tid + 256 * it;
    if (i >= PXB * 9) break;
    const int r = i / 9, tap = i - r * 9;
    const int m = m0 + r;
    f32x4 dA = {0.0f, 0.0f, 0.0f, 0.0f}, dB = {0.0f, 0.0f, 0.0f, 0.0f};
    if (m < p.M) {
      const int b = m / HW, rem = m - b * HW;
      const int ho = rem / p.W, wo = rem - ho * p.W;
      const int ti = tap / 3, tj = tap - ti * 3;
      const float hf = (float)(ho - 1 + ti) + omy[it];
      const float wf = (float)(wo - 1 + tj) + omx[it];
      const bool inside = hf > -1.0f && hf < (float)p.H && wf > -1.0f && wf < (float)p.W;
      const float hfl = floorf(hf), wfl = floorf(wf);
      const int hl = inside ? (int)hfl : 0, wl = inside ? (int)wfl : 0;
      const float lh = hf - hfl, lw = wf - wfl, hh = 1.0f - lh, hw = 1.0f - lw;
      const bool t_ok = inside && hl >= 0, b_ok = inside && hl + 1 <= p.H - 1;
      const bool l_ok = wl >= 0, r_ok = wl + 1 <= p.W - 1;
      const int y0 = max(hl, 0), x0 = max(wl, 0);
      const int y1 = min(hl + 1, p.H - 1), x1 = min(wl + 1, p.W - 1);     // (hl + 1 >= 0 whenever inside)
      dA[0] = __int_as_float(((b * p.H + y0) * p.W + x0) * p.C);
      dA[1] = __int_as_float((max(x1, x0) - x0) * p.C);
      dA[2] = __int_as_float((max(y1, y0) - y0) * p.W * p.C);
      dA[3] = (p.mask_activated ? omm[it] : cf_sigmoid(omm[it])) * p.in_scale;
      dB[0] = (t_ok && l_ok) ? hh * hw : 0.0f;
      dB[1] = (t_ok && r_ok) ? hh * lw : 0.0f;
      dB[2] = (b_ok && l_ok) ? lh * hw : 0.0f;
      dB[3] = (b_ok && r_ok) ? lh * lw : 0.0f;
      // a corner that is clamped away shares its address with a valid one, so its weight must be zero:
      // true by construction (x1 == x0 only if !l_ok or !r_ok; y1 == y0 only if !t_ok or !b_ok)
    }
    desc[2 * i] = dA;
    desc[2 * i + 1] = dB;
  }
  __syncthreads();
  DPROF_MARK(0)

  // corner samples are requested TWO chunks ahead (sets c & 1): the texture path, which bounds the
  // 64-channel layers, then always has a full chunk of requests queued behind the one being blended
  // (DEEP only for the two-pixel-group configuration: with WP = 1 the second set costs an occupancy
  //  step or spills and measured slower)
  // (... and for the half-size tiles, CT = 1: one set keeps them at 114 registers = FOUR workgroups per CU, which beats the
  //  deeper queue at three: 8 x 64 -> 64 at 112 x 200 106-108 vs 115-116 us, step 7.53 vs 7.63 ms.  CF_DCN_DEEP1: dev A/B)
#ifdef CF_DCN_DEEP1
  constexpr bool DEEP = WP == 2;
#else
  constexpr bool DEEP = WP == 2 && CT == 2;
#endif
  constexpr int NSET = DEEP ? 2 : 1;
  f32x4 cvs[NSET][NP][4][2];   // 4 corners x 8 channels
  f32x4 cws[NSET][NP];         // corner weights
  float cmks[NSET][NP];        // 16 * sigmoid(mask)
  auto load_b_pair = [&](int c, int i, f32x4 (&cv)[NP][4][2], f32x4 (&cw)[NP], float (&cmk)[NP]) __attribute__((always_inline)) {
    const int tap = c / p.chunks_per_tap;
    const int c0 = (c - tap * p.chunks_per_tap) * 32 + (tid & 3) * 8;
    {
      const int e = (((tid + 256 * i) >> 2) * 9 + tap) * 2;
      const f32x4 dA = desc[e];
      cw[i] = desc[e + 1];
      cmk[i] = dA[3];
#ifdef CF_DCN_NOGATHER    // (dev timing experiment: every corner from one line - what the kernel costs without the gather)
      const float* a0 = p.x + ((__float_as_int(dA[0]) & 0) + c0);
#else
      const float* a0 = p.x + (__float_as_int(dA[0]) + c0);
#endif
      const float* a1 = a0 + __float_as_int(dA[1]);
      const float* a2 = a0 + __float_as_int(dA[2]);
      const float* a3 = a2 + __float_as_int(dA[1]);
#ifdef CF_DCN_NOLOAD      // (dev timing experiment: no corner requests at all behind the first chunk of a tile)
      if (c > 1) return;
#endif
      cv[i][0][0] = *reinterpret_cast<const f32x4*>(a0);
      cv[i][0][1] = *reinterpret_cast<const f32x4*>(a0 + 4);
      cv[i][1][0] = *reinterpret_cast<const f32x4*>(a1);
      cv[i][1][1] = *reinterpret_cast<const f32x4*>(a1 + 4);
      cv[i][2][0] = *reinterpret_cast<const f32x4*>(a2);
      cv[i][2][1] = *reinterpret_cast<const f32x4*>(a2 + 4);
      cv[i][3][0] = *reinterpret_cast<const f32x4*>(a3);
      cv[i][3][1] = *reinterpret_cast<const f32x4*>(a3 + 4);
    }
  };
  auto load_b = [&](int c, f32x4 (&cv)[NP][4][2], f32x4 (&cw)[NP], float (&cmk)[NP]) {
#pragma unroll
    for (int i = 0; i < NP; ++i) load_b_pair(c, i, cv, cw, cmk);
  };
  auto store_b = [&](unsigned char* buf, const f32x4 (&cv)[NP][4][2], const f32x4 (&cw)[NP], const float (&cmk)[NP]) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int pr = tid + 256 * i;
      const float mk = cmk[i];   // the mask (x 2^4 activation scale) is applied after the 4-corner sum, as the reference does
#ifdef CF_DCN_NOBLEND       // (dev timing experiment: no blend / split arithmetic, the first corner's bits are staged as they are)
      {
        unsigned char* o = buf + (pr >> 2) * FROWB + (pr & 3) * 16;
        *reinterpret_cast<f32x4*>(o) = cv[i][0][0] * mk;
        *reinterpret_cast<f32x4*>(o + PLANE) = cv[i][0][1];
        continue;
      }
#endif
      // explicit vector FMAs (v_pk_fma_f32: two channels per instruction)
      f32x4 v0 = cw[i][0] * cv[i][0][0], v1 = cw[i][0] * cv[i][0][1];
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        const f32x4 wk = {cw[i][k], cw[i][k], cw[i][k], cw[i][k]};
        v0 = __builtin_elementwise_fma(wk, cv[i][k][0], v0);
        v1 = __builtin_elementwise_fma(wk, cv[i][k][1], v1);
      }
      v0 *= mk;
      v1 *= mk;
      u32x4 hi, lo;                 // (the activation scale is already in mk)
      { unsigned th, tl; split2(v0[0], v0[1], th, tl); hi[0] = th; lo[0] = tl; }
      { unsigned th, tl; split2(v0[2], v0[3], th, tl); hi[1] = th; lo[1] = tl; }
      { unsigned th, tl; split2(v1[0], v1[1], th, tl); hi[2] = th; lo[2] = tl; }
      { unsigned th, tl; split2(v1[2], v1[3], th, tl); hi[3] = th; lo[3] = tl; }
      unsigned char* o = buf + (pr >> 2) * FROWB + (pr & 3) * 16;
      *reinterpret_cast<u32x4*>(o) = hi;
      *reinterpret_cast<u32x4*>(o + PLANE) = lo;
    }
  };

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
  f16x8 wh[2][RT], wl[2][RT];
  auto load_w = [&](f16x8 (&dh)[RT], f16x8 (&dl)[RT], int ks) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      dh[rt] = *wfrag16(p.weight, w_ok ? rt0 + rt : 0, ks, 0, n_ks, lane);
      dl[rt] = *wfrag16(p.weight, w_ok ? rt0 + rt : 0, ks, 1, n_ks, lane);
    }
  };
  auto mma_kstep = [&](const unsigned char* buf, int s, const f16x8 (&ah)[RT], const f16x8 (&al)[RT]) {
    f16x8 xh[CT], xl[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const unsigned char* row = buf + (wp * 32 * CT + ct * 32 + li) * FROWB + s * 32 + h * 16;
      xh[ct] = *reinterpret_cast<const f16x8*>(row);
      xl[ct] = *reinterpret_cast<const f16x8*>(row + PLANE);
    }
#ifdef CF_DCN_NOMFMA       // (dev timing experiment: ONE MFMA per k-step that still consumes every operand register)
    {
      f16x8 a = al[0] + ah[0], b = xh[0] + xl[0] + xh[CT - 1] + xl[CT - 1];
#pragma unroll
      for (int rt = 1; rt < RT; ++rt) a += al[rt] + ah[rt];
      accm[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, accm[0][0], 0, 0, 0);
      return;
    }
#endif
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        accs[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[rt], xh[ct], accs[rt][ct], 0, 0, 0);
        accs[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[rt], xl[ct], accs[rt][ct], 0, 0, 0);
        accm[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[rt], xh[ct], accm[rt][ct], 0, 0, 0);
      }
  };

  // K split over gridDim.z (small maps with long K: 88 tiles x 144 chunks at 14x25 cannot fill the chip,
  // and one tile's chunk chain is latency-bound): this workgroup owns chunks [c_lo, c_hi)
  const int c_lo = (int)((long)p.n_chunks * blockIdx.z / gridDim.z);
  const int c_hi = (int)((long)p.n_chunks * (blockIdx.z + 1) / gridDim.z);
  // chunk j (relative to c_lo) lives in LDS buffer j & 1 and (DEEP) in register set j & 1
  const int n_own = c_hi - c_lo;
  load_b(c_lo, cvs[0], cws[0], cmks[0]);
  load_w(wh[0], wl[0], 2 * c_lo);
  load_w(wh[1], wl[1], 2 * c_lo + 1);
  if (DEEP && n_own > 1) load_b(c_lo + 1, cvs[NSET - 1], cws[NSET - 1], cmks[NSET - 1]);
  store_b(smem, cvs[0], cws[0], cmks[0]);
  if (n_own > NSET) load_b(c_lo + NSET, cvs[0], cws[0], cmks[0]);
  __syncthreads();
  // (an earlier form with exec-masked corner loads inside a pinned loop glitched when launched behind unrelated kernels,
  //  tools/stress_dcn.py: the loads below are unconditional)
  auto iteration = [&](int j, f32x4 (&cv)[NP][4][2], f32x4 (&cw)[NP], float (&cmk)[NP]) {
    // MFMAs of chunk j; then chunk j+1 (held in set cv) is blended into the other buffer and the set is
    // re-requested for chunk j+1+NSET
    unsigned char* cur = smem + (j & 1) * BUF;
    unsigned char* nxt = smem + ((j + 1) & 1) * BUF;
    const int c = c_lo + j;
    // Straight-line, hand-interleaved form for the single-pixel-group tiles (WP == 1: the 128- / 256-channel layers on the
    // 28 x 50 and 14 x 25 maps): every MFMA is followed by one PIECE of the next chunk's staging (blend of 8 channels x 4
    // corners, operand split + LDS store, the corner requests of the chunk after that) and a sched_barrier keeps it
    // there, so the MFMA executes while the wave issues the piece; indices are clamped instead of branching and every
    // load is unconditional (no exec-masked operand load inside a pinned loop).  Bit-identical to
    // the plain form.  Measured (tools/bench_dcn.py, same box): 256 -> 128 at 28 x 50: 86.4 vs 93.0 us; the two-group tiles
    // of the 64-channel layers LOSE with it (254 vs 212 us, 149 vs 131 us: every wait for a corner or an LDS fragment then
    // also holds back the wave's next MFMA), so they keep the compiler's order.  CF_DCN_NOPIN: dev A/B.
#ifndef CF_DCN_NOPIN
    if constexpr (WP == 1 && CT == 2)
    {
      f32x4 bv[2];                           // blended 8 channels of the pair in progress
      u32x4 bhi, blo;
      auto work = [&](int slot) __attribute__((always_inline)) {
        constexpr int NSTG = 4 * WP;         // staging pieces: 4 per (pixel, unit) pair
        if (slot < NSTG) {
          const int i = slot >> 2, part = slot & 3;
          if (part < 2) {                    // blend: 4 channels... x2 (one f32x4 half of the 8-channel unit), then the mask
            f32x4 v = cw[i][0] * cv[i][0][part];
#pragma unroll
            for (int k = 1; k < 4; ++k) {
              const f32x4 wk4 = {cw[i][k], cw[i][k], cw[i][k], cw[i][k]};
              v = __builtin_elementwise_fma(wk4, cv[i][k][part], v);
            }
            bv[part] = v * cmk[i];
          } else {                           // split to fp16 hi / lo; the second half also stores the unit
            const int hf = part - 2;
            { unsigned th, tl; split2(bv[hf][0], bv[hf][1], th, tl); bhi[2 * hf] = th; blo[2 * hf] = tl; }
            { unsigned th, tl; split2(bv[hf][2], bv[hf][3], th, tl); bhi[2 * hf + 1] = th; blo[2 * hf + 1] = tl; }
            if (hf == 1) {
              const int pr = tid + 256 * i;
              unsigned char* o = nxt + (pr >> 2) * FROWB + (pr & 3) * 16;
              *reinterpret_cast<u32x4*>(o) = bhi;
              *reinterpret_cast<u32x4*>(o + PLANE) = blo;
            }
          }
        } else if (slot < NSTG + WP) {       // corner requests of chunk c + 1 + NSET into the set just consumed
          load_b_pair(min(c + 1 + NSET, c_hi - 1), slot - NSTG, cv, cw, cmk);
        }
      };
      auto kstep = [&](int s, const f16x8 (&ah)[RT], const f16x8 (&al)[RT]) __attribute__((always_inline)) {
        f16x8 xh[CT], xl[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const unsigned char* row = cur + (wp * 32 * CT + ct * 32 + li) * FROWB + s * 32 + h * 16;
          xh[ct] = *reinterpret_cast<const f16x8*>(row);
          xl[ct] = *reinterpret_cast<const f16x8*>(row + PLANE);
        }
        int slot = s * 6 * RT;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            accs[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[rt], xh[ct], accs[rt][ct], 0, 0, 0);
            work(slot++);
            __builtin_amdgcn_sched_barrier(0);
            accs[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[rt], xl[ct], accs[rt][ct], 0, 0, 0);
            work(slot++);
            __builtin_amdgcn_sched_barrier(0);
            accm[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[rt], xh[ct], accm[rt][ct], 0, 0, 0);
            work(slot++);
            __builtin_amdgcn_sched_barrier(0);
          }
      };
      kstep(0, wh[0], wl[0]);
      load_w(wh[0], wl[0], min(2 * c + 2, 2 * c_hi - 2));
      __builtin_amdgcn_sched_barrier(0);
      kstep(1, wh[1], wl[1]);
      load_w(wh[1], wl[1], min(2 * c + 3, 2 * c_hi - 1));
      __syncthreads();
      return;
    }
#endif
    mma_kstep(cur, 0, wh[0], wl[0]);
#ifndef CF_DCN_NOWEIGHT   // (dev timing experiment: the weight stream's share of the texture path - DESIGN.md section 9)
    if (j + 1 < n_own) load_w(wh[0], wl[0], 2 * c + 2);
#endif
    mma_kstep(cur, 1, wh[1], wl[1]);
    if (j + 1 < n_own) {
#ifndef CF_DCN_NOWEIGHT
      load_w(wh[1], wl[1], 2 * c + 3);
#endif
      store_b(nxt, cv, cw, cmk);
      if (j + 1 + NSET < n_own) load_b(c + 1 + NSET, cv, cw, cmk);
    }
#ifndef CF_DCN_NOBARRIER   // (dev timing experiment: what the per-chunk workgroup barrier costs; results are garbage)
    __syncthreads();
#endif
  };
  if (DEEP) {
    for (int j = 0; j < n_own; j += 2) {
      iteration(j, cvs[NSET - 1], cws[NSET - 1], cmks[NSET - 1]);     // chunk j+1 was requested into set 1
      if (j + 1 < n_own) iteration(j + 1, cvs[0], cws[0], cmks[0]);
    }
  } else {
    for (int j = 0; j < n_own; ++j) iteration(j, cvs[0], cws[0], cmks[0]);
  }

  DPROF_MARK(1)
  if (gridDim.z > 1) {   // raw partial sums; scale / bias / activation happen in the reduction
    const int ns = p.n_rt * 32;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int m = m0 + wp * 32 * CT + ct * 32 + li;
      if (m >= p.M || !w_ok) continue;
      float* o = p.partial + ((size_t)blockIdx.z * p.M + m) * ns;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = accm[rt][ct][g * 4 + e] + accs[rt][ct][g * 4 + e];
          *reinterpret_cast<f32x4*>(o + (rt0 + rt) * 32 + 8 * g + 4 * h) = v;
        }
    }
    return;
  }

  // Coalesced epilogue (as in cf_conv3x3_f16.hip): each wave transposes its 32 pixels x 32*RT channels through a private
  // LDS tile (free after the loop's last barrier) and writes whole pixel rows - RT*128 contiguous bytes per pixel
  // instead of 32-byte pieces - and the split-bf16 copy as 8-byte pieces that are contiguous across lanes.
  constexpr bool coalesced = COAL;        // (the host selects it: N % 4 == 0)
#ifdef CF_DCN_NOEPI         // (dev timing experiment: no output transposition / stores - one conditional store keeps the sums alive)
  {
    float sum = 0.0f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += accm[rt][ct][r] + accs[rt][ct][r];
    if (sum == 12345.0f) p.out[tid] = sum;
    return;
  }
#endif
  if (coalesced && w_ok) {
    constexpr int EROW = RT * 128 + 16;
    constexpr int LPP = RT * 8, PPI = 64 / LPP;
    asm volatile("; cf_epilogue_begin" ::: "memory");   // marker for tools/check_isa.py (no instruction)
    unsigned char* eb = smem + wave * 32 * EROW;
    const int chunk = lane % LPP, psub = lane / LPP;
    const int n = rt0 * 32 + chunk * 4;
    const bool n_ok = n < p.N;
    f32x4 bias4 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (n_ok) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
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
        const size_t m = (size_t)m0 + wp * 32 * CT + ct * 32 + ploc;
        f32x4 v = *reinterpret_cast<const f32x4*>(eb + ploc * EROW + chunk * 16) + bias4;
        if (n_ok && m < (size_t)p.M) {
          if (p.act == CF_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
          }
          *reinterpret_cast<f32x4*>(p.out + m * p.out_stride + n) = v;
          if (p.out_split) {   // hi = rne_bf16(v), lo = rne_bf16(v - hi): the head kernels' input format
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const __bf16 h0 = (__bf16)v[2 * e], h1 = (__bf16)v[2 * e + 1];
              const __bf16 l0 = (__bf16)(v[2 * e] - (float)h0), l1 = (__bf16)(v[2 * e + 1] - (float)h1);
              w[e] = ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16) | __builtin_bit_cast(unsigned short, h0);
              w[2 + e] = ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16) | __builtin_bit_cast(unsigned short, l0);
            }
            unsigned* o = p.out_split + (m * 2 * p.split_stride + n) / 2;
            *reinterpret_cast<uint2*>(o) = uint2{w[0], w[1]};
            *reinterpret_cast<uint2*>(o + p.split_stride / 2) = uint2{w[2], w[3]};
          }
        }
      }
      if constexpr (RT == 1) {
        // the mx rows of the heads (cf_head_fused mx = 1) from the same tile: this wave holds one 32-channel block of its 32
        // pixels; lane l < 32 packs pixel l exactly as cf_pack_feat_mx would from `out` (same fp32 values: tile + bias, ReLU)
        if (p.out_mx && lane < 32) {
          const size_t m = (size_t)m0 + wp * 32 * CT + ct * 32 + lane;
          if (m < (size_t)p.M) {
            float v[32];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              f32x4 t = *reinterpret_cast<const f32x4*>(eb + lane * EROW + i * 16) + *reinterpret_cast<const f32x4*>(p.bias + rt0 * 32 + 4 * i);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[4 * i + e] = p.act == CF_ACT_RELU ? fmaxf(t[e], 0.0f) : t[e];
            }
            mx_pack_block(v, p.out_mx + m * 272, rt0, p.mx_scale);
          }
        }
      }
    }
  }

#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int m = m0 + wp * 32 * CT + ct * 32 + li;
    if (coalesced) break;
    if (m >= p.M) continue;
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
          if (p.act == CF_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
          }
          *reinterpret_cast<f32x4*>(p.out + (size_t)m * p.out_stride + n) = v;
          if (p.out_split) {   // hi = rne_bf16(v), lo = rne_bf16(v - hi): the head kernels' input format
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const __bf16 h0 = (__bf16)v[2 * e], h1 = (__bf16)v[2 * e + 1];
              const __bf16 l0 = (__bf16)(v[2 * e] - (float)h0), l1 = (__bf16)(v[2 * e + 1] - (float)h1);
              w[e] = ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16) | __builtin_bit_cast(unsigned short, h0);
              w[2 + e] = ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16) | __builtin_bit_cast(unsigned short, l0);
            }
            unsigned* o = p.out_split + ((size_t)m * 2 * p.split_stride + n) / 2;
            *reinterpret_cast<uint2*>(o) = uint2{w[0], w[1]};
            *reinterpret_cast<uint2*>(o + p.split_stride / 2) = uint2{w[2], w[3]};
          }
        } else {
          for (int e = 0; e < 4 && n + e < p.N; ++e) {
            float x = v[e] + p.bias[n + e];
            if (p.act == CF_ACT_RELU) x = fmaxf(x, 0.0f);
            p.out[(size_t)m * p.out_stride + n + e] = x;
          }
        }
      }
  }
#ifdef CF_DCN_PROF
  DPROF_MARK(2)
  if (tid == 0 && blockIdx.y == 0 && m0 < p.M)
    for (int i = 0; i < 3; ++i) p.out[(size_t)m0 * p.out_stride + i] = (float)t_prof[i];
#endif
}

// K-split reduction: out = act((sum_z partial[z]) * out_scale + bias), partials added in z order
__global__ __launch_bounds__(256) void dcn_reduce_kernel(const float* __restrict__ partial, int ks, long MN4, int ns4,
                                                         int N, const float* __restrict__ bias, float out_scale,
                                                         int act, float* __restrict__ out, int out_stride) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < MN4; i += (long)gridDim.x * 256) {
    const long m = i / ns4;
    const int n = (int)(i - m * ns4) * 4;
    if (n >= N) continue;
    f32x4 v = reinterpret_cast<const f32x4*>(partial)[i];
    for (int z = 1; z < ks; ++z) v += reinterpret_cast<const f32x4*>(partial)[(size_t)z * MN4 + i];
    v = v * out_scale;
    for (int e = 0; e < 4 && n + e < N; ++e) {
      float x = v[e] + bias[n + e];
      if (act == CF_ACT_RELU) x = fmaxf(x, 0.0f);
      out[(size_t)m * out_stride + n + e] = x;
    }
  }
}

// number of K parts of a DCN launch on an H x W map (per-image geometry only)
int dcn_k_split(int H, int W, int n_chunks, int n_pad) {
  const long hw = (long)H * W;
  int ks = hw <= 512 ? 4 : (hw <= 2048 && n_pad <= 128) ? 2 : 1;   // (256 outputs at 28x50: the reduction pass costs what the split saves)
  while (ks > 1 && n_chunks < 4 * ks) ks >>= 1;
  return ks;
}

template <typename K, typename A>
void launch_f16(K kernel, dim3 grid, size_t dyn, hipStream_t st, const A& args) {
  static CfLdsLimit lds_limit;  // one per template instantiation
  lds_limit.ensure(kernel, dyn, 16384);
  hipLaunchKernelGGL(kernel, grid, dim3(256), dyn, st, args);
}

}  // namespace

extern "C" int cf_conv2d_f16x3(const cf_conv_args* a, void* stream) {
  CF_REQUIRE(a != nullptr, "cf_conv2d_f16x3: null args");
  CF_REQUIRE(a->n_src >= 1 && a->n_src <= CF_MAX_SRC, "cf_conv2d_f16x3: n_src=%d", a->n_src);
  CF_REQUIRE(a->K_pad > 0 && a->K_pad % 32 == 0, "cf_conv2d_f16x3: K_pad=%d not a multiple of 32", a->K_pad);
  CF_REQUIRE(a->N > 0 && a->N_pad >= a->N && a->N_pad % 32 == 0, "cf_conv2d_f16x3: N=%d N_pad=%d", a->N, a->N_pad);
  CF_REQUIRE(a->N_pad == 32 || a->N_pad % 64 == 0, "cf_conv2d_f16x3: N_pad=%d must be 32 or a multiple of 64", a->N_pad);
  CF_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->Ho > 0 && a->Wo > 0 && a->stride > 0, "cf_conv2d_f16x3: bad geometry");
  CF_REQUIRE(a->weight && a->slots && a->bias && a->out, "cf_conv2d_f16x3: null buffer");
  CF_REQUIRE(a->out_layout == CF_LAYOUT_NHWC && a->out_stride >= a->N && a->out_stride % 4 == 0,
             "cf_conv2d_f16x3: output must be fp32 NHWC with a stride that is a multiple of 4");
  CF_REQUIRE(a->act == CF_ACT_NONE || a->act == CF_ACT_RELU, "cf_conv2d_f16x3: act=%d unsupported", a->act);
  CF_REQUIRE(a->out_scale > 0.0f, "cf_conv2d_f16x3: out_scale must be the 2^-(s+4) the weights were packed with");
  CF_REQUIRE(!a->residual || a->res_stride % 4 == 0, "cf_conv2d_f16x3: residual stride must be a multiple of 4");
  for (int i = 0; i < a->n_src; ++i)
    CF_REQUIRE(a->src[i] && a->src_c[i] > 0 && a->src_c[i] % 8 == 0, "cf_conv2d_f16x3: source %d invalid", i);
  const long M = (long)a->B * a->Ho * a->Wo;
  CF_REQUIRE((long)a->B * a->H * a->W < (1L << 30) && M < (1L << 31), "cf_conv2d_f16x3: tensor too large");
  ConvF k{};
  for (int i = 0; i < CF_MAX_SRC; ++i) {
    k.src[i] = i < a->n_src ? a->src[i] : nullptr;
    k.src_c[i] = i < a->n_src ? a->src_c[i] : 0;
  }
  k.weight = reinterpret_cast<const unsigned char*>(a->weight);
  k.slots = a->slots;
  k.bias = a->bias;
  k.residual = a->residual;
  k.out = a->out;
  k.H = a->H; k.W = a->W; k.Ho = a->Ho; k.Wo = a->Wo; k.stride = a->stride;
  k.n_chunks = a->K_pad / 32;
  k.res_stride = a->res_stride; k.out_stride = a->out_stride; k.act = a->act;
  k.M = (int)M; k.N = a->N; k.HoWo = a->Ho * a->Wo;
  k.n_rt = a->N_pad / 32;
  k.out_scale = a->out_scale;
  k.in_scale = cf_resolve_in_scale(a->in_scale);
  CF_REQUIRE(k.in_scale > 0.0f, "cf_conv2d_f16x3: in_scale must be 0 (= 16) or a power of two");
  const size_t dyn = (size_t)k.n_chunks * 4 * sizeof(cf_slot);
  hipStream_t st = (hipStream_t)stream;
  if (a->N_pad == 32) {
    launch_f16(conv_f16x3_kernel<1, 4, 1, false>, dim3((unsigned)((M + 255) / 256), 1), dyn, st, k);
  } else if (a->N_pad == 64) {
    launch_f16(conv_f16x3_kernel<1, 4, 2, false>, dim3((unsigned)((M + 255) / 256), 1), dyn, st, k);
  } else if (a->N_pad == 128) {
    launch_f16(conv_f16x3_kernel<2, 2, 2, true>, dim3((unsigned)((M + 127) / 128), 1), dyn, st, k);
  } else {
    launch_f16(conv_f16x3_kernel<4, 1, 2, true>, dim3((unsigned)((M + 63) / 64), (unsigned)((a->N_pad + 255) / 256)), dyn, st, k);
  }
  return cf_check_launch("cf_conv2d_f16x3");
}

extern "C" int cf_dcn_v2_f16x3(const cf_dcn_args* a, void* stream) {
  CF_REQUIRE(a != nullptr, "cf_dcn_v2_f16x3: null args");
  CF_REQUIRE(a->C > 0 && a->C % 32 == 0, "cf_dcn_v2_f16x3: C=%d not a multiple of 32", a->C);
  CF_REQUIRE(a->N > 0 && a->N_pad >= a->N && a->N_pad % 32 == 0, "cf_dcn_v2_f16x3: N=%d N_pad=%d", a->N, a->N_pad);
  CF_REQUIRE(a->N_pad <= 128 || a->N_pad % 64 == 0, "cf_dcn_v2_f16x3: N_pad=%d above 128 must be a multiple of 64 (two row tiles per wave)", a->N_pad);
  CF_REQUIRE(a->om_stride >= 27, "cf_dcn_v2_f16x3: om_stride=%d < 27", a->om_stride);
  CF_REQUIRE(a->x && a->offmask && a->weight && a->bias && a->out, "cf_dcn_v2_f16x3: null buffer");
  CF_REQUIRE(a->out_stride >= a->N && a->out_stride % 4 == 0, "cf_dcn_v2_f16x3: bad out_stride");
  CF_REQUIRE(a->act == CF_ACT_NONE || a->act == CF_ACT_RELU, "cf_dcn_v2_f16x3: act=%d unsupported", a->act);
  CF_REQUIRE(a->out_scale > 0.0f, "cf_dcn_v2_f16x3: out_scale missing");
  const long M = (long)a->B * a->H * a->W;
  CF_REQUIRE(M > 0 && M * a->C < (1L << 31), "cf_dcn_v2_f16x3: bad geometry / tensor too large");
  DcnF k{};
  k.x = a->x; k.om = a->offmask; k.weight = reinterpret_cast<const unsigned char*>(a->weight);
  k.bias = a->bias; k.out = a->out;
  k.om_stride = a->om_stride; k.H = a->H; k.W = a->W; k.C = a->C;
  k.n_chunks = 9 * a->C / 32;
  k.chunks_per_tap = a->C / 32;
  k.out_stride = a->out_stride; k.act = a->act; k.M = (int)M; k.N = a->N;
  k.n_rt = a->N_pad / 32;
  k.out_scale = a->out_scale;
  k.in_scale = cf_resolve_in_scale(a->in_scale);
  k.mx_scale = cf_resolve_in_scale(a->mx_scale);
  CF_REQUIRE(k.in_scale > 0.0f && k.mx_scale > 0.0f, "cf_dcn_v2_f16x3: in_scale / mx_scale must be 0 (= 16) or a power of two");
  k.out_split = static_cast<unsigned*>(a->out_split_bf16);
  k.split_stride = a->split_stride;
  k.out_mx = static_cast<unsigned char*>(a->out_mx);
  CF_REQUIRE(!a->out_mx || (a->N == 64 && a->N_pad == 64 && (a->N & 3) == 0),
             "cf_dcn_v2_f16x3: the mx output is the 64-channel feature map's (N = N_pad = 64)");
  CF_REQUIRE(!a->out_split_bf16 || (a->split_stride >= a->N && a->split_stride % 8 == 0 && a->N % 4 == 0),
             "cf_dcn_v2_f16x3: split output needs N %% 4 == 0 and a plane stride >= N that is a multiple of 8");
  hipStream_t st = (hipStream_t)stream;
  // K split for small maps, when the caller provides the workspace (decided per image geometry, never
  // by the batch size: it changes the summation order, and a shard has to reproduce the full batch)
  const unsigned ks = a->workspace ? (unsigned)dcn_k_split(a->H, a->W, k.n_chunks, a->N_pad) : 1u;
  CF_REQUIRE(ks == 1 || (!a->out_split_bf16 && !a->out_mx), "cf_dcn_v2_f16x3: the split-bf16 / mx outputs are not available on K-split maps");
  CF_REQUIRE(ks == 1 || a->workspace_bytes >= (size_t)ks * M * a->N_pad * sizeof(float),
             "cf_dcn_v2_f16x3: workspace of %zu bytes is smaller than cf_dcn_v2_workspace_bytes(...)", a->workspace_bytes);
  k.partial = static_cast<float*>(a->workspace);
  k.mask_activated = a->mask_activated;
  static const int direct_epi = [] { const char* e = getenv("CF_DCN_EPI"); return e ? atoi(e) == 0 : 0; }();
  k.direct_epilogue = direct_epi;
  const bool coal = (a->N & 3) == 0 && !k.direct_epilogue;   // whole-row epilogue through LDS
  CF_REQUIRE(!a->out_mx || coal, "cf_dcn_v2_f16x3: the mx output is written by the whole-row epilogue (CF_DCN_EPI=0 disables it)");
  // SMALL GRIDS (small batches): 64 output channels on 64-pixel tiles (the 128-channel configuration with two of its
  // four channel-group waves idle in the MFMAs, all four staging) while that launch still fits the chip in one round:
  // 128 -> 64 at 56x100, bs=1: 30.4 vs 43.4 us, bs=2: 31.6 vs 45.0 us; at 350 workgroups the gain is gone.  Same K order,
  // same K split: bit-identical, so the choice may depend on the batch size (as in cf_conv3x3_f16x3).
#ifdef CF_DCN_SMALLTILE     // (dev timing experiment: the 64-pixel-tile configuration at every grid size)
  if (a->N_pad <= 64) {
#else
  if (a->N_pad <= 64 && (M + 63) / 64 * (long)ks <= 256) {
#endif
    const dim3 grid((unsigned)((M + 63) / 64), 1u, ks);
    if (coal) launch_f16(dcn_f16x3_kernel<4, 1, 1, true>, grid, 0, st, k);
    else launch_f16(dcn_f16x3_kernel<4, 1, 1, false>, grid, 0, st, k);
  } else if (a->N_pad <= 64) {          // 64 channels: 2 x 32-channel wave rows, 2 x 64 pixels
    // half-size pixel tiles (32 pixels per wave: 114 registers, 39 KB of LDS - four workgroups per CU instead of two): the parts
    // of this kernel add up instead of overlapping (docs/experiments/r5_dcn_attribution.md), so a third wave per SIMD pays
    // wherever the grid is not many rounds deep - 8 x 64 -> 64 at 112 x 200: 116.5 vs 123.5 us, 8 x 128 -> 64 at 56 x 100: 61.1 vs
    // 72.6 us, 16 x 64 -> 64 at 112 x 200: equal; step 7.75 vs 7.80 ms.  Same K order: bit-identical.  CF_DCN_CT1=0: dev A/B.
    static const int ct1 = [] { const char* e = getenv("CF_DCN_CT1"); return e ? atoi(e) : 1; }();
    const dim3 grid((unsigned)((M + 127) / 128), (unsigned)((a->N_pad + 63) / 64), ks);
    const dim3 grid1((unsigned)((M + 63) / 64), (unsigned)((a->N_pad + 63) / 64), ks);
    if (ct1 && coal) launch_f16(dcn_f16x3_kernel<2, 2, 1, true, 1>, grid1, 0, st, k);
    else if (coal) launch_f16(dcn_f16x3_kernel<2, 2, 1, true>, grid, 0, st, k);
    else launch_f16(dcn_f16x3_kernel<2, 2, 1, false>, grid, 0, st, k);
  } else if (a->N_pad <= 128) {  // 128 channels: 4 x 32-channel wave rows, 64 pixels
    const dim3 grid((unsigned)((M + 63) / 64), (unsigned)((a->N_pad + 127) / 128), ks);
    if (coal) launch_f16(dcn_f16x3_kernel<4, 1, 1, true>, grid, 0, st, k);
    else launch_f16(dcn_f16x3_kernel<4, 1, 1, false>, grid, 0, st, k);
  } else {
    const dim3 grid((unsigned)((M + 63) / 64), (unsigned)((a->N_pad + 255) / 256), ks);
    if (coal) launch_f16(dcn_f16x3_kernel<4, 1, 2, true>, grid, 0, st, k);
    else launch_f16(dcn_f16x3_kernel<4, 1, 2, false>, grid, 0, st, k);
  }
  if (ks > 1) {
    const int ns4 = k.n_rt * 32 / 4;
    const long MN4 = M * ns4;
    const long blocks = (MN4 + 255) / 256;
    hipLaunchKernelGGL(dcn_reduce_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st, k.partial,
                       (int)ks, MN4, ns4, a->N, a->bias, a->out_scale, a->act, a->out, a->out_stride);
  }
  return cf_check_launch("cf_dcn_v2_f16x3");
}

extern "C" size_t cf_dcn_v2_workspace_bytes(int B, int H, int W, int C, int N_pad) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || N_pad <= 0) return 0;
  const int ks = dcn_k_split(H, W, 9 * C / 32, N_pad);
  return ks > 1 ? (size_t)ks * B * H * W * N_pad * sizeof(float) : 0;
}
