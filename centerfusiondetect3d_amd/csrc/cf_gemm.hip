// Implicit-GEMM convolution and DCNv2 on the gfx950 fp32 MFMA pipe (v_mfma_f32_32x32x2_f32).
//
// GEMM view  D[m][n] = sum_k A[m][k] * Wt[n][k]
//   m : output pixel (b, ho, wo)          M = B*Ho*Wo
//   n : output channel                     N_pad (multiple of 32)
//   k : (tap, input channel) in host-defined slot order, K_pad multiple of 32
// A is never materialised: each workgroup stages a BM x 32 A-chunk into LDS straight from the
// NHWC activations (conv: plain taps; DCN: 4-corner bilinear gather, mask-modulated), and a
// BN x 32 chunk of the pre-packed weights.  Four waves (256 threads) per workgroup; each wave owns
// a (TM*32) x (TN*32) sub-tile as TM*TN 32x32 fp32 accumulators (16 VGPRs each).
//
// LDS rows are padded to 36 floats so the per-lane ds_read_b128 of 16 consecutive rows lands on 16
// distinct 4-bank slots (conflict-free); all 64 lanes of a wave read row (lane&31), k-offset
// 4*(lane>>5) - the two halves of the wave supply the two k-slots of each 32x32x2 MFMA.
//
// fp32 MFMA is exact fp32 (an fmaf chain) and runs at 64 FLOP/clk/SIMD, so one workgroup spends
// 64 cycles per MFMA: global->LDS staging (one float4 per thread per 32 rows) hides completely
// behind it once the next chunk is prefetched into registers during the current chunk's MFMAs.
#include "cf_common.h"

namespace {

struct EpilogueArgs {
  const float* bias;
  const float* residual;
  float* out;
  float* out2;
  int res_stride, out_stride, out_layout, act;
  int M, N, HoWo;
};

struct ConvK {
  const float* src[CF_MAX_SRC];
  int src_c[CF_MAX_SRC];
  const float* weight;
  const cf_slot* slots;
  int H, W, Ho, Wo, stride, K_pad, n_chunks, NT;
  EpilogueArgs ep;
};

struct DcnK {
  const float* x;
  const float* om;
  const float* weight;
  int om_stride, H, W, C, K_pad, n_chunks, NT, chunks_per_tap;
  int mask_activated;   // offmask channels 18..26 are modulation factors already (no sigmoid here)
  EpilogueArgs ep;
};

// One 32-deep K chunk.  PRECISE: the chunk is summed into a fresh accumulator (first MFMA takes
// C = 0) and only then added to the running sum, i.e. two-level (blocked) summation: rounding error
// grows with sqrt(32) + sqrt(K/32) instead of sqrt(K).  A bare fp32 MFMA chain over K = 4608 is
// ~4x less accurate than the CPU reference's blocked accumulation, and the DCN neck amplifies
// upstream rounding ~100x (tools/stage_error.py), so everything feeding the neck runs PRECISE.
template <int TM, int TN, bool PRECISE>
__device__ __forceinline__ void mma_chunk(const float* __restrict__ As, const float* __restrict__ Bs,
                                          int a_row0, int b_row0, int lane, f32x16 (&acc)[TM][TN]) {
  const int li = lane & 31, h = lane >> 5;
  f32x16 part[TM][TN];
#pragma unroll
  for (int ks = 0; ks < CF_BK / 8; ++ks) {
    f32x4 a[TM], b[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
      a[tm] = *reinterpret_cast<const f32x4*>(&As[(a_row0 + tm * 32 + li) * CF_LDS_STRIDE + ks * 8 + h * 4]);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
      b[tn] = *reinterpret_cast<const f32x4*>(&Bs[(b_row0 + tn * 32 + li) * CF_LDS_STRIDE + ks * 8 + h * 4]);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          if (PRECISE) {
            if (ks == 0 && t == 0) {
              const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
              part[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][t], b[tn][t], zero, 0, 0, 0);
            } else {
              part[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][t], b[tn][t], part[tm][tn], 0, 0, 0);
            }
          } else {
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][t], b[tn][t], acc[tm][tn], 0, 0, 0);
          }
        }
  }
  if (PRECISE) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] += part[tm][tn];
  }
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == CF_ACT_RELU) return fmaxf(v, 0.0f);
  if (act == CF_ACT_SIGMOID_CLAMP) return fminf(fmaxf(cf_sigmoid(v), 1e-4f), 1.0f - 1e-4f);
  return v;
}

// C/D layout of the 32x32 accumulator: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
template <int TM, int TN>
__device__ __forceinline__ void epilogue(const EpilogueArgs& ep, int m_base, int n_base, int lane,
                                         f32x16 (&acc)[TM][TN]) {
  const int li = lane & 31, h = lane >> 5;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n_base + tn * 32 + li;
    const bool n_ok = n < ep.N;
    const float bias = n_ok ? ep.bias[n] : 0.0f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int m4 = m_base + tm * 32 + 8 * g + 4 * h;  // 4 consecutive pixels m4..m4+3
        if (ep.out_layout == CF_LAYOUT_NHWC) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int m = m4 + e;
            if (n_ok && m < ep.M) {
              float v = acc[tm][tn][g * 4 + e] + bias;
              if (ep.residual) v += ep.residual[(size_t)m * ep.res_stride + n];
              ep.out[(size_t)m * ep.out_stride + n] = apply_act(v, ep.act);
            }
          }
        } else {  // NCHW: lanes differ in channel, the 4 accumulator rows are 4 consecutive pixels
          if (n_ok && m4 < ep.M) {
            const int b = m4 / ep.HoWo, pix = m4 - b * ep.HoWo;
            const size_t o = ((size_t)b * ep.N + n) * ep.HoWo + pix;
            f32x4 v, v2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float raw = acc[tm][tn][g * 4 + e] + bias;
              v[e] = apply_act(raw, ep.act);
              v2[e] = 0.0f;
              if (ep.act == CF_ACT_RAW_AND_SIGDEPTH) v2[e] = 1.0f / (cf_sigmoid(raw) + 1e-6f) - 1.0f;
            }
            if (m4 + 3 < ep.M && pix + 3 < ep.HoWo && (o & 3) == 0) {
              *reinterpret_cast<f32x4*>(&ep.out[o]) = v;
              if (ep.act == CF_ACT_RAW_AND_SIGDEPTH) *reinterpret_cast<f32x4*>(&ep.out2[o]) = v2;
            } else {
              for (int e = 0; e < 4; ++e) {
                const int m = m4 + e;
                if (m < ep.M) {
                  const int bb = m / ep.HoWo, pp = m - bb * ep.HoWo;
                  const size_t oo = ((size_t)bb * ep.N + n) * ep.HoWo + pp;
                  ep.out[oo] = v[e];
                  if (ep.act == CF_ACT_RAW_AND_SIGDEPTH) ep.out2[oo] = v2[e];
                }
              }
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Convolution
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N, bool PRECISE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvK p) {
  constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
  constexpr int RA = BM / 32, RB = BN / 32;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * CF_LDS_STRIDE];
  extern __shared__ __attribute__((aligned(16))) cf_slot lds_slots[];  // the layer's slot table
  float* As = smem;
  float* Bs = smem + BM * CF_LDS_STRIDE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lid = cf_xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid / p.NT, nt = lid - mt * p.NT;
  const int m0 = mt * BM, n0 = nt * BN;
  const int tr = tid >> 3, ts = tid & 7;
  for (int i = tid; i < p.n_chunks * 8; i += 256) lds_slots[i] = p.slots[i];
  __syncthreads();

  // per-thread staging rows: pixel -> top-left input coordinate
  int y0[RA], x0[RA], boff[RA];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int m = m0 + tr + 32 * j;
    if (m < p.ep.M) {
      const int b = m / HoWo, rem = m - b * HoWo;
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

  f32x4 ra[RA], rb[RB];
  auto load_chunk = [&](int c) {
    const cf_slot sl = lds_slots[c * 8 + ts];
    const int src = __builtin_amdgcn_readfirstlane(lds_slots[c * 8].src);
    const float* sp = src == 1 ? p.src[1] : src == 2 ? p.src[2] : src == 3 ? p.src[3] : p.src[0];
    const int sc = src == 1 ? p.src_c[1] : src == 2 ? p.src_c[2] : src == 3 ? p.src_c[3] : p.src_c[0];
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const int y = y0[j] + sl.dy, x = x0[j] + sl.dx;
      const bool ok = (src >= 0) && (sl.c_off >= 0) && ((unsigned)y < (unsigned)p.H) &&
                      ((unsigned)x < (unsigned)p.W);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(sp + (size_t)(boff[j] + y * p.W + x) * sc + sl.c_off);
      ra[j] = v;
    }
    const float* wp = p.weight + (size_t)(n0 + tr) * p.K_pad + c * CF_BK + ts * 4;
#pragma unroll
    for (int j = 0; j < RB; ++j) rb[j] = *reinterpret_cast<const f32x4*>(wp + (size_t)(32 * j) * p.K_pad);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  load_chunk(0);
  for (int c = 0; c < p.n_chunks; ++c) {
    __syncthreads();  // previous chunk's LDS reads are done
#pragma unroll
    for (int j = 0; j < RA; ++j)
      *reinterpret_cast<f32x4*>(&As[(tr + 32 * j) * CF_LDS_STRIDE + ts * 4]) = ra[j];
#pragma unroll
    for (int j = 0; j < RB; ++j)
      *reinterpret_cast<f32x4*>(&Bs[(tr + 32 * j) * CF_LDS_STRIDE + ts * 4]) = rb[j];
    __syncthreads();
    if (c + 1 < p.n_chunks) load_chunk(c + 1);  // in flight while the MFMAs below run
    mma_chunk<TM, TN, PRECISE>(As, Bs, wm * TM * 32, wn * TN * 32, lane, acc);
  }
  epilogue<TM, TN>(p.ep, m0 + wm * TM * 32, n0 + wn * TN * 32, lane, acc);
}

// ---------------------------------------------------------------------------------------------
// Convolution with <= 16 output channels (stem 7x7 3->16, level0 3x3 16->16 at full resolution):
// v_mfma_f32_16x16x4_f32 tiles so no MFMA work is spent on padding N to 32.  128 pixels x 16
// channels per workgroup; each wave owns 32 pixel rows (two 16x16 accumulators of 4 VGPRs).
// Lane l = (i = l & 15, q = l >> 4) reads 4 consecutive k (one ds_read_b128) of row i and feeds
// element e to MFMA e, whose 4 k-slots are then {16g + 4q + e | q = 0..3}.
// ---------------------------------------------------------------------------------------------
template <bool PRECISE>
__global__ __launch_bounds__(256) void conv_igemm_n16_kernel(ConvK p) {
  constexpr int BM = 128, RA = 4;
  __shared__ __attribute__((aligned(16))) float smem[(BM + 16) * CF_LDS_STRIDE];
  extern __shared__ __attribute__((aligned(16))) cf_slot lds_slots[];
  float* As = smem;
  float* Bs = smem + BM * CF_LDS_STRIDE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM;
  const int tr = tid >> 3, ts = tid & 7;
  for (int i = tid; i < p.n_chunks * 8; i += 256) lds_slots[i] = p.slots[i];
  __syncthreads();

  int y0[RA], x0[RA], boff[RA];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int m = m0 + tr + 32 * j;
    if (m < p.ep.M) {
      const int b = m / HoWo, rem = m - b * HoWo;
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

  f32x4 ra[RA], rb;
  auto load_chunk = [&](int c) {
    const cf_slot sl = lds_slots[c * 8 + ts];
    const int src = __builtin_amdgcn_readfirstlane(lds_slots[c * 8].src);
    const float* sp = src == 1 ? p.src[1] : src == 2 ? p.src[2] : src == 3 ? p.src[3] : p.src[0];
    const int sc = src == 1 ? p.src_c[1] : src == 2 ? p.src_c[2] : src == 3 ? p.src_c[3] : p.src_c[0];
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const int y = y0[j] + sl.dy, x = x0[j] + sl.dx;
      const bool ok = (src >= 0) && (sl.c_off >= 0) && ((unsigned)y < (unsigned)p.H) &&
                      ((unsigned)x < (unsigned)p.W);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(sp + (size_t)(boff[j] + y * p.W + x) * sc + sl.c_off);
      ra[j] = v;
    }
    if (tid < 128) rb = *reinterpret_cast<const f32x4*>(p.weight + (size_t)tr * p.K_pad + c * CF_BK + ts * 4);
  };

  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const int i16 = lane & 15, q = lane >> 4;
  load_chunk(0);
  for (int c = 0; c < p.n_chunks; ++c) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RA; ++j)
      *reinterpret_cast<f32x4*>(&As[(tr + 32 * j) * CF_LDS_STRIDE + ts * 4]) = ra[j];
    if (tid < 128) *reinterpret_cast<f32x4*>(&Bs[tr * CF_LDS_STRIDE + ts * 4]) = rb;
    __syncthreads();
    if (c + 1 < p.n_chunks) load_chunk(c + 1);
    f32x4 part[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(&Bs[i16 * CF_LDS_STRIDE + g * 16 + q * 4]);
      f32x4 a[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
        a[t] = *reinterpret_cast<const f32x4*>(&As[(wave * 32 + t * 16 + i16) * CF_LDS_STRIDE + g * 16 + q * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (PRECISE) part[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][e], b[e], part[t], 0, 0, 0);
          else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][e], b[e], acc[t], 0, 0, 0);
        }
    }
    if (PRECISE) {
      acc[0] += part[0];
      acc[1] += part[1];
    }
  }
  // C/D layout of the 16x16 tile: col = lane & 15 (channel), row = 4 * (lane >> 4) + reg (pixel)
  const int n = i16;
  if (n < p.ep.N) {
    const float bias = p.ep.bias[n];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wave * 32 + t * 16 + 4 * q + r;
        if (m < p.ep.M) {
          float v = acc[t][r] + bias;
          if (p.ep.residual) v += p.ep.residual[(size_t)m * p.ep.res_stride + n];
          p.ep.out[(size_t)m * p.ep.out_stride + n] = apply_act(v, p.ep.act);
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------
// DCNv2 (3x3, stride 1, pad 1): bilinear gather fused into the A-chunk staging
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N, bool PRECISE>
__global__ __launch_bounds__(256) void dcn_igemm_kernel(DcnK p) {
  constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
  constexpr int RA = BM / 32, RB = BN / 32;
  // LDS: A tile | B tile | sampling descriptors {h, w, sigmoid(mask), -} per (tile row, tap)
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * CF_LDS_STRIDE + BM * 9 * 4];
  float* As = smem;
  float* Bs = smem + BM * CF_LDS_STRIDE;
  f32x4* desc = reinterpret_cast<f32x4*>(smem + (BM + BN) * CF_LDS_STRIDE);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lid = cf_xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid / p.NT, nt = lid - mt * p.NT;
  const int m0 = mt * BM, n0 = nt * BN;
  const int tr = tid >> 3, ts = tid & 7;
  const int HW = p.H * p.W;

  // ---- once per tile: sampling position and modulation of every (pixel, tap).  The offsets,
  // the floor/weights and above all the sigmoid are shared by all channel chunks of a tap.
  for (int i = tid; i < BM * 9; i += 256) {
    const int r = i / 9, tap = i - r * 9;
    const int m = m0 + r;
    f32x4 d = {-1.0e9f, -1.0e9f, 0.0f, 0.0f};
    if (m < p.ep.M) {
      const int b = m / HW, rem = m - b * HW;
      const int ho = rem / p.W, wo = rem - ho * p.W;
      const float* om = p.om + (size_t)m * p.om_stride;
      const int ti = tap / 3, tj = tap - ti * 3;
      d[0] = (float)(ho - 1 + ti) + om[2 * tap];
      d[1] = (float)(wo - 1 + tj) + om[2 * tap + 1];
      d[2] = p.mask_activated ? om[18 + tap] : cf_sigmoid(om[18 + tap]);
    }
    desc[i] = d;
  }

  int boff[RA];
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int m = m0 + tr + 32 * j;
    boff[j] = (m < p.ep.M ? m / HW : 0) * HW;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  __syncthreads();

  // Software pipeline: the four corner rows of chunk c+1 (and its weight chunk) are requested
  // before the MFMAs of chunk c and combined after them.
  f32x4 cv[RA][4], cw[RA], rb[RB];
  auto issue = [&](int c) {
    const int tap = c / p.chunks_per_tap;
    const int c0 = (c - tap * p.chunks_per_tap) * CF_BK + ts * 4;
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const f32x4 d = desc[(tr + 32 * j) * 9 + tap];
      const float hf = d[0], wf = d[1];
      const bool inside = hf > -1.0f && hf < (float)p.H && wf > -1.0f && wf < (float)p.W;
      const float hfl = floorf(hf), wfl = floorf(wf);
      const int hl = (int)hfl, wl = (int)wfl;
      const float lh = hf - hfl, lw = wf - wfl, hh = 1.0f - lh, hw = 1.0f - lw;
      const bool t_ok = inside && hl >= 0, b_ok = inside && hl + 1 <= p.H - 1;
      const bool l_ok = wl >= 0, r_ok = wl + 1 <= p.W - 1;
      const float* base = p.x + (size_t)boff[j] * p.C + c0;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      cv[j][0] = (t_ok && l_ok) ? *reinterpret_cast<const f32x4*>(base + (size_t)(hl * p.W + wl) * p.C) : z;
      cv[j][1] = (t_ok && r_ok) ? *reinterpret_cast<const f32x4*>(base + (size_t)(hl * p.W + wl + 1) * p.C) : z;
      cv[j][2] = (b_ok && l_ok) ? *reinterpret_cast<const f32x4*>(base + (size_t)((hl + 1) * p.W + wl) * p.C) : z;
      cv[j][3] = (b_ok && r_ok) ? *reinterpret_cast<const f32x4*>(base + (size_t)((hl + 1) * p.W + wl + 1) * p.C) : z;
      const f32x4 w = {hh * hw, hh * lw, lh * hw, lh * lw};
      cw[j] = w;  // the mask is applied after the 4-corner sum, as the reference does
    }
    const float* wp = p.weight + (size_t)(n0 + tr) * p.K_pad + c * CF_BK + ts * 4;
#pragma unroll
    for (int j = 0; j < RB; ++j) rb[j] = *reinterpret_cast<const f32x4*>(wp + (size_t)(32 * j) * p.K_pad);
  };

  issue(0);
  for (int c = 0; c < p.n_chunks; ++c) {
    const int tap = c / p.chunks_per_tap;
    f32x4 ra[RA];
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const float mk = desc[(tr + 32 * j) * 9 + tap][2];
      ra[j] = (cw[j][0] * cv[j][0] + cw[j][1] * cv[j][1] + cw[j][2] * cv[j][2] + cw[j][3] * cv[j][3]) * mk;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RA; ++j)
      *reinterpret_cast<f32x4*>(&As[(tr + 32 * j) * CF_LDS_STRIDE + ts * 4]) = ra[j];
#pragma unroll
    for (int j = 0; j < RB; ++j)
      *reinterpret_cast<f32x4*>(&Bs[(tr + 32 * j) * CF_LDS_STRIDE + ts * 4]) = rb[j];
    __syncthreads();
    if (c + 1 < p.n_chunks) issue(c + 1);
    mma_chunk<TM, TN, PRECISE>(As, Bs, wm * TM * 32, wn * TN * 32, lane, acc);
  }
  epilogue<TM, TN>(p.ep, m0 + wm * TM * 32, n0 + wn * TN * 32, lane, acc);
}

struct TileCfg {
  int bm, bn;
};

// Launch with `dyn` bytes of dynamic LDS on top of the kernel's static tiles (raises the kernel's
// dynamic-LDS limit once per instantiation; 160 KiB per workgroup are available on gfx950).
template <typename K, typename A>
void launch_dyn(K kernel, int blocks, size_t dyn, hipStream_t st, const A& args) {
  static CfLdsLimit lds_limit;  // one per template instantiation
  lds_limit.ensure(kernel, dyn, 32768);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), dyn, st, args);
}

// Tile choice: widest N tile that divides N_pad; 128-row tiles only when the grid still gives every
// CU (256) at least two workgroups; and when even 64-row tiles leave some CUs with one workgroup
// while their neighbours hold two (small-M layers: level5, 14x25 / 28x50 neck nodes), halve the N tile.
TileCfg pick_tile(long M, int N_pad) {
  TileCfg t;
  t.bn = (N_pad % 128 == 0) ? 128 : (N_pad % 64 == 0) ? 64 : 32;
  const long blocks128 = ((M + 127) / 128) * (N_pad / t.bn);
  t.bm = blocks128 >= 512 ? 128 : 64;
  if (t.bm == 64 && t.bn == 128 && ((M + 63) / 64) * (N_pad / 128) < 512) t.bn = 64;
  return t;
}

}  // namespace

#define LAUNCH_TILE(KERNEL, ARGS, BM_, BN_, WM_, WN_)                                   \
  do {                                                                                  \
    const int MT = (int)((M + BM_ - 1) / BM_), NT = N_pad / BN_;                        \
    ARGS.NT = NT;                                                                       \
    if (precise) launch_dyn((KERNEL<BM_, BN_, WM_, WN_, true>), MT * NT, dyn_lds, st, ARGS);    \
    else launch_dyn((KERNEL<BM_, BN_, WM_, WN_, false>), MT * NT, dyn_lds, st, ARGS);           \
  } while (0)

#define DISPATCH_TILE(KERNEL, ARGS, FORCE_BM64)                            \
  do {                                                                     \
    TileCfg t = pick_tile(M, N_pad);                                       \
    if ((FORCE_BM64) && t.bn >= 64) t.bm = 64;                             \
    if (t.bm == 128 && t.bn == 128) LAUNCH_TILE(KERNEL, ARGS, 128, 128, 2, 2); \
    else if (t.bm == 128 && t.bn == 64) LAUNCH_TILE(KERNEL, ARGS, 128, 64, 2, 2); \
    else if (t.bm == 128 && t.bn == 32) LAUNCH_TILE(KERNEL, ARGS, 128, 32, 4, 1); \
    else if (t.bm == 64 && t.bn == 128) LAUNCH_TILE(KERNEL, ARGS, 64, 128, 1, 4); \
    else if (t.bm == 64 && t.bn == 64) LAUNCH_TILE(KERNEL, ARGS, 64, 64, 2, 2);   \
    else LAUNCH_TILE(KERNEL, ARGS, 128, 32, 4, 1);                         \
  } while (0)

extern "C" int cf_conv2d_fused(const cf_conv_args* a, void* stream) {
  CF_REQUIRE(a != nullptr, "cf_conv2d_fused: null args");
  CF_REQUIRE(a->n_src >= 1 && a->n_src <= CF_MAX_SRC, "cf_conv2d_fused: n_src=%d", a->n_src);
  CF_REQUIRE(a->K_pad > 0 && a->K_pad % CF_BK == 0, "cf_conv2d_fused: K_pad=%d not a multiple of 32", a->K_pad);
  CF_REQUIRE(a->N_pad >= a->N && a->N_pad % 32 == 0 && a->N > 0, "cf_conv2d_fused: N=%d N_pad=%d", a->N, a->N_pad);
  CF_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->Ho > 0 && a->Wo > 0 && a->stride > 0, "cf_conv2d_fused: bad geometry");
  CF_REQUIRE(a->weight && a->slots && a->bias && a->out, "cf_conv2d_fused: null buffer");
  CF_REQUIRE(a->act != CF_ACT_RAW_AND_SIGDEPTH || (a->out2 && a->out_layout == CF_LAYOUT_NCHW),
             "cf_conv2d_fused: RAW_AND_SIGDEPTH needs out2 and NCHW");
  CF_REQUIRE(a->out_layout == CF_LAYOUT_NCHW || a->out_stride >= a->N, "cf_conv2d_fused: out_stride < N");
  for (int i = 0; i < a->n_src; ++i)
    CF_REQUIRE(a->src[i] && a->src_c[i] > 0 && a->src_c[i] % 4 == 0, "cf_conv2d_fused: source %d invalid", i);
  const long M = (long)a->B * a->Ho * a->Wo;
  CF_REQUIRE((long)a->B * a->H * a->W * 4 < (1L << 31) && M < (1L << 31), "cf_conv2d_fused: tensor too large");
  ConvK k{};
  for (int i = 0; i < CF_MAX_SRC; ++i) {
    k.src[i] = i < a->n_src ? a->src[i] : nullptr;
    k.src_c[i] = i < a->n_src ? a->src_c[i] : 0;
  }
  k.weight = a->weight;
  k.slots = a->slots;
  k.H = a->H; k.W = a->W; k.Ho = a->Ho; k.Wo = a->Wo; k.stride = a->stride;
  k.K_pad = a->K_pad;
  k.n_chunks = a->K_pad / CF_BK;
  k.ep = EpilogueArgs{a->bias, a->residual, a->out, a->out2, a->res_stride, a->out_stride,
                      a->out_layout, a->act, (int)M, a->N, a->Ho * a->Wo};
  const int N_pad = a->N_pad;
  const bool precise = a->precise != 0;
  const size_t dyn_lds = (size_t)(a->K_pad / 4) * sizeof(cf_slot);
  hipStream_t st = (hipStream_t)stream;
  if (a->N <= 16 && a->out_layout == CF_LAYOUT_NHWC && a->act != CF_ACT_SIGMOID_CLAMP) {
    const int blocks = (int)((M + 127) / 128);
    if (precise) launch_dyn(conv_igemm_n16_kernel<true>, blocks, dyn_lds, st, k);
    else launch_dyn(conv_igemm_n16_kernel<false>, blocks, dyn_lds, st, k);
    return cf_check_launch("cf_conv2d_fused");
  }
  // PRECISE doubles the accumulator registers: 64-row tiles keep three workgroups per CU
  DISPATCH_TILE(conv_igemm_kernel, k, precise);
  return cf_check_launch("cf_conv2d_fused");
}

extern "C" int cf_dcn_v2_fused(const cf_dcn_args* a, void* stream) {
  CF_REQUIRE(a != nullptr, "cf_dcn_v2_fused: null args");
  CF_REQUIRE(a->C > 0 && a->C % CF_BK == 0, "cf_dcn_v2_fused: C=%d not a multiple of 32", a->C);
  CF_REQUIRE(a->N_pad >= a->N && a->N_pad % 32 == 0 && a->N > 0, "cf_dcn_v2_fused: N=%d N_pad=%d", a->N, a->N_pad);
  CF_REQUIRE(a->om_stride >= 27, "cf_dcn_v2_fused: om_stride=%d < 27", a->om_stride);
  CF_REQUIRE(a->x && a->offmask && a->weight && a->bias && a->out, "cf_dcn_v2_fused: null buffer");
  CF_REQUIRE(a->out_stride >= a->N, "cf_dcn_v2_fused: out_stride < N");
  const long M = (long)a->B * a->H * a->W;
  CF_REQUIRE(M > 0 && M * a->C < (1L << 31) * 2, "cf_dcn_v2_fused: bad geometry");
  DcnK k{};
  k.x = a->x; k.om = a->offmask; k.weight = a->weight;
  k.om_stride = a->om_stride; k.H = a->H; k.W = a->W; k.C = a->C;
  k.K_pad = 9 * a->C;
  k.n_chunks = k.K_pad / CF_BK;
  k.chunks_per_tap = a->C / CF_BK;
  k.mask_activated = a->mask_activated;
  k.ep = EpilogueArgs{a->bias, nullptr, a->out, nullptr, 0, a->out_stride, CF_LAYOUT_NHWC, a->act,
                      (int)M, a->N, a->H * a->W};
  const int N_pad = a->N_pad;
  const bool precise = a->precise != 0;
  const size_t dyn_lds = 0;
  hipStream_t st = (hipStream_t)stream;
  // 64-row tiles: the pipelined gather holds 4 corner rows per staged pixel row in registers
  DISPATCH_TILE(dcn_igemm_kernel, k, true);
  return cf_check_launch("cf_dcn_v2_fused");
}
