// Fused tail of a detection head on the bf16 MFMA pipe (split operands, see cf_gemm_bf16.hip):
//
//     x (256 hidden channels of a 64-pixel tile)  ->  [ReLU(W_l x + b_l)] x n_hidden  ->  W_out x + b_out
//
// replaces the per-layer launches of model/networks/detectHeads.py:80-90 (1x1 256->256 + ReLU
// layers) and :64-71 (1x1 256->n_out) whose only HBM-visible result is the small NCHW head map.
// The hidden maps never return to HBM: one workgroup keeps its pixel tile [64 px][256 ch] (hi and lo
// bf16 planes, 66 KiB) in LDS for the whole chain.
//
// GEMM orientation is SWAPPED with respect to the conv kernels: MFMA A-operand = weights
// (rows = output channels), B-operand = activations (columns = pixels).  Consequences:
//   * weights never touch LDS: they are pre-packed on the host in MFMA fragment order, so a wave's
//     A fragment of one 16-deep k-step is ONE fully coalesced 1 KiB global load (L2-resident);
//   * no barrier inside a layer - only one between layers, when the tile is rewritten in place;
//   * the accumulator has pixels on lanes, so the final NCHW store is coalesced along pixels.
// Wave w of the 4 owns output channels [64w, 64w+64) x all 64 pixels (2x2 32x32 accumulators); in
// the output layer the 4 waves split K instead and their partial sums are reduced through LDS.
#include <stdlib.h>
#include "cf_common.h"
#include "cf_mx.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 hf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 hf16x32 __attribute__((ext_vector_type(32)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x6 __attribute__((ext_vector_type(6)));

constexpr int HT_PX = 64;            // pixels per workgroup
constexpr int HT_C = 256;            // hidden width
constexpr int HT_ROWB = HT_C * 2 + 16;  // LDS bytes per pixel row per plane (528: odd multiple of 16)
constexpr int HT_PLANE = HT_PX * HT_ROWB;
constexpr int HT_LDS = 2 * HT_PLANE;    // 67,584 B

struct HeadTailK {
  const unsigned char* x;   // split-bf16 NHWC [M][2][x_stride]
  int x_stride;
  int M, HW;
  int n_heads;
  int n_hidden;             // 256->256 layers per head (0..2)
  const unsigned char* w_hidden[CF_MAX_HEADS][2];  // fragment-packed 256x256
  const float* b_hidden[CF_MAX_HEADS][2];          // 256 floats
  const unsigned char* w_out[CF_MAX_HEADS];        // fragment-packed 32x256
  const float* b_out[CF_MAX_HEADS];                // 32 floats (padded)
  float* out[CF_MAX_HEADS];                        // NCHW fp32 (B, n_out, H, W)
  float* out2[CF_MAX_HEADS];                       // RAW_AND_SIGDEPTH second output or null
  int c_base[CF_MAX_HEADS];                        // first hidden channel of the head inside x
  int n_out[CF_MAX_HEADS];
  int act[CF_MAX_HEADS];
};

__device__ __forceinline__ float bf16_rne(float a) { return (float)(__bf16)a; }
__device__ __forceinline__ unsigned pack2(float a, float b) {
  return ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)b) << 16) |
         __builtin_bit_cast(unsigned short, (__bf16)a);
}

// fragment-packed weights: [row tile][k step][plane][lane][8 bf16]  -> byte offset of a wave's fragment
__device__ __forceinline__ const bf16x8* wfrag(const unsigned char* w, int rt, int ks, int plane, int n_ks, int lane) {
  // (tile base) + 16 * lane: with a wave-uniform rt / ks the base is scalar arithmetic (see wfrag16, cf_f16x3.h)
  const unsigned char* base = w + ((size_t)rt * n_ks + ks) * 2048 + plane * 1024;
  return reinterpret_cast<const bf16x8*>(base + (unsigned)lane * 16u);
}

// accumulator (lane = pixel ct*32+li, reg r = channel 64w + 32rt + (r&3) + 8(r>>2) + 4h) -> ReLU(acc + b)
// -> split bf16 -> LDS tile [plane][px][528 B]
#ifdef CF_LEGACY_HEADS   // (32x32x16 / slot-table head kernels of rounds 1-3: nothing dispatches them in the default build)
__device__ __forceinline__ void store_hidden_tile(unsigned char* xt, const f32x16 (&acc)[2][2], const float* bias,
                                                  int wave, int li, int h) {
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = wave * 64 + rt * 32 + 8 * g + 4 * h;
        const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + ch);
        float v[4], hi[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = fmaxf(acc[rt][ct][g * 4 + e] + bb[e], 0.0f);
          hi[e] = bf16_rne(v[e]);
        }
        unsigned char* o = xt + (ct * 32 + li) * HT_ROWB + ch * 2;
        const u32x2 ph = {pack2(hi[0], hi[1]), pack2(hi[2], hi[3])};
        const u32x2 pl = {pack2(v[0] - hi[0], v[1] - hi[1]), pack2(v[2] - hi[2], v[3] - hi[3])};
        *reinterpret_cast<u32x2*>(o) = ph;
        *reinterpret_cast<u32x2*>(o + HT_PLANE) = pl;
      }
}

// the same from 16x16 accumulators (v_mfma_f32_16x16x32_bf16: lane = pixel ct*16 + (l & 15), reg r = channel
// 64w + 16rt + 4(l >> 4) + r) of one 64-pixel half tile
#endif  // CF_LEGACY_HEADS

template <bool SCALED = false>
__device__ __forceinline__ void store_hidden_tile16(unsigned char* xt, const f32x4 (&acc)[4][4], const float* bias,
                                                    int wave, int lane, float sc = 1.0f, int pxcol = -1) {
  const int g = lane >> 4, c16 = pxcol < 0 ? (lane & 15) : pxcol;   // pxcol: the pixel column this lane's accumulators hold
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const int ch = wave * 64 + rt * 16 + 4 * g;
    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + ch);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      float v[4], hi[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = fmaxf(SCALED ? __builtin_fmaf(acc[rt][ct][e], sc, bb[e]) : acc[rt][ct][e] + bb[e], 0.0f);
        hi[e] = bf16_rne(v[e]);
      }
      unsigned char* o = xt + (ct * 16 + c16) * HT_ROWB + ch * 2;
      const u32x2 ph = {pack2(hi[0], hi[1]), pack2(hi[2], hi[3])};
      const u32x2 pl = {pack2(v[0] - hi[0], v[1] - hi[1]), pack2(v[2] - hi[2], v[3] - hi[3])};
      *reinterpret_cast<u32x2*>(o) = ph;
      *reinterpret_cast<u32x2*>(o + HT_PLANE) = pl;
    }
  }
}

// pixel index inside a 64-pixel tile -> (image, pixel inside the image); false = outside
struct FlatMap {       // 64 consecutive pixels of the flattened (B*H*W) index space
  int m0, M, HW;
  __device__ __forceinline__ bool operator()(int px, int& b, int& pix) const {
    const int m = m0 + px;
    if (m >= M) return false;
    b = m / HW;
    pix = m - b * HW;
    return true;
  }
};
struct TileMap {       // 64 pixels of a tile at (y0, x0) of image b: 4 rows of 16 (sh = 4) or 8 rows of 8 (sh = 3)
  int b, y0, x0, H, W, sh = 4;
  __device__ __forceinline__ bool operator()(int px, int& bb, int& pix) const {
    const int y = y0 + (px >> sh), x = x0 + (px & ((1 << sh) - 1));
    if (y >= H || x >= W) return false;
    bb = b;
    pix = y * W + x;
    return true;
  }
};

// 4 per-wave partial output tiles [32 n][64 px] -> LDS -> sum + bias + activation -> NCHW
#ifdef CF_LEGACY_HEADS   // (32x32x16 / slot-table head kernels of rounds 1-3: nothing dispatches them in the default build)
template <class PixMap>
__device__ __forceinline__ void reduce_and_store(const HeadTailK& p, unsigned char* xt, const f32x16 (&oacc)[2],
                                                 int head, const PixMap& pm) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  float* red = reinterpret_cast<float*>(xt);
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = (r & 3) + 8 * (r >> 2) + 4 * h;
      red[(wave * 32 + n) * HT_PX + ct * 32 + li] = oacc[ct][r];
    }
  __syncthreads();
  const int n_out = p.n_out[head], act = p.act[head];
  const float* bo = p.b_out[head];
  float* out = p.out[head];
  float* out2 = p.out2[head];
  const int px = tid & 63;
  int b, pix;
  if (pm(px, b, pix)) {
    for (int n = tid >> 6; n < n_out; n += 4) {
      const float raw = red[n * HT_PX + px] + red[(32 + n) * HT_PX + px] + red[(64 + n) * HT_PX + px] +
                        red[(96 + n) * HT_PX + px] + bo[n];
      const size_t o = ((size_t)b * n_out + n) * p.HW + pix;
      float v = raw;
      if (act == CF_ACT_RELU) v = fmaxf(raw, 0.0f);
      else if (act == CF_ACT_SIGMOID_CLAMP) v = fminf(fmaxf(cf_sigmoid(raw), 1e-4f), 1.0f - 1e-4f);
      out[o] = v;
      if (act == CF_ACT_RAW_AND_SIGDEPTH) out2[o] = 1.0f / (cf_sigmoid(raw) + 1e-6f) - 1.0f;
    }
  }
}

// Hidden layers + output layer on a pixel tile that is already in LDS (xt).  All 256 threads.
template <class PixMap>
__device__ __forceinline__ void head_tail_from_lds(const HeadTailK& p, unsigned char* xt, int head, const PixMap& pm) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;

  // ---- hidden layers: acc[rt][ct] = W[64w + 32rt .. +32][:] . X[:][32ct .. +32]
  for (int l = 0; l < p.n_hidden; ++l) {
    const unsigned char* w = p.w_hidden[head][l];
    const float* bias = p.b_hidden[head][l];
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    bf16x8 wh[3][2], wl[3][2];  // [set = ks % 3][rt]: three k-steps ahead
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        wh[t][rt] = *wfrag(w, wave * 2 + rt, t, 0, 16, lane);
        wl[t][rt] = *wfrag(w, wave * 2 + rt, t, 1, 16, lane);
      }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int cur = ks % 3;
      bf16x8 xh[2], xl[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const unsigned char* row = xt + (ct * 32 + li) * HT_ROWB + (ks * 16 + h * 8) * 2;
        xh[ct] = *reinterpret_cast<const bf16x8*>(row);
        xl[ct] = *reinterpret_cast<const bf16x8*>(row + HT_PLANE);
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[cur][rt], xh[ct], acc[rt][ct], 0, 0, 0);
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cur][rt], xl[ct], acc[rt][ct], 0, 0, 0);
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cur][rt], xh[ct], acc[rt][ct], 0, 0, 0);
        }
      if (ks + 3 < 16) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          wh[cur][rt] = *wfrag(w, wave * 2 + rt, ks + 3, 0, 16, lane);
          wl[cur][rt] = *wfrag(w, wave * 2 + rt, ks + 3, 1, 16, lane);
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the look-ahead loads from sinking to their first use
    }
    __syncthreads();  // every wave has read the whole tile: rewrite it in place
    store_hidden_tile(xt, acc, bias, wave, li, h);
    __syncthreads();
  }

  // ---- output layer: out[n][px] = sum_k Wout[n][k] X[k][px]; wave w takes k in [64w, 64w+64)
  f32x16 oacc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[b][r] = 0.0f;
  {
    const unsigned char* w = p.w_out[head];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ks = wave * 4 + s;
      const bf16x8 ah = *wfrag(w, 0, ks, 0, 16, lane), al = *wfrag(w, 0, ks, 1, 16, lane);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const unsigned char* row = xt + (ct * 32 + li) * HT_ROWB + (ks * 16 + h * 8) * 2;
        const bf16x8 xh = *reinterpret_cast<const bf16x8*>(row);
        const bf16x8 xl = *reinterpret_cast<const bf16x8*>(row + HT_PLANE);
        oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh, oacc[ct], 0, 0, 0);
        oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl, oacc[ct], 0, 0, 0);
        oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh, oacc[ct], 0, 0, 0);
      }
    }
  }
  __syncthreads();  // tile no longer needed: reuse LDS for the 4 partial sums [wave][n 32][px 64]
  reduce_and_store(p, xt, oacc, head, pm);
}

// The same chain on v_mfma_f32_16x16x32_bf16 (weights packed by pack_fragments16: w_hidden [16 rt][8 ks], w_out one
// 16-row tile): wave w owns channels [64w, 64w+64) x 64 pixels as 4 x 4 accumulators of 16 x 16; a k-step is 32 deep.
#endif  // CF_LEGACY_HEADS

template <class PixMap>
__device__ __forceinline__ void head_tail_from_lds16(const HeadTailK& p, unsigned char* xt, int head, const PixMap& pm) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c16 = lane & 15;
  for (int l = 0; l < p.n_hidden; ++l) {
    const unsigned char* w = p.w_hidden[head][l];
    const float* bias = p.b_hidden[head][l];
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.0f;
    bf16x8 wh[2][4], wl[2][4];               // one k-step ahead: set ks & 1
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      wh[0][rt] = *wfrag(w, wave * 4 + rt, 0, 0, 8, lane);
      wl[0][rt] = *wfrag(w, wave * 4 + rt, 0, 1, 8, lane);
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks + 1 < 8) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          wh[(ks + 1) & 1][rt] = *wfrag(w, wave * 4 + rt, ks + 1, 0, 8, lane);
          wl[(ks + 1) & 1][rt] = *wfrag(w, wave * 4 + rt, ks + 1, 1, 8, lane);
        }
      }
      bf16x8 xh[4], xl[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const unsigned char* row = xt + (ct * 16 + c16) * HT_ROWB + (ks * 32 + g * 8) * 2;
        xh[ct] = *reinterpret_cast<const bf16x8*>(row);
        xl[ct] = *reinterpret_cast<const bf16x8*>(row + HT_PLANE);
      }
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ks & 1][rt], xh[ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks & 1][rt], xl[ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks & 1][rt], xh[ct], acc[rt][ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();  // every wave has read the whole tile: rewrite it in place
    store_hidden_tile16(xt, acc, bias, wave, lane);
    __syncthreads();
  }

  // ---- output layer: wave w takes k in [64w, 64w+64) = 2 k-steps
  f32x4 oacc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) oacc[b][r] = 0.0f;
  {
    const unsigned char* w = p.w_out[head];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int ks = wave * 2 + s2;
      const bf16x8 ah = *wfrag(w, 0, ks, 0, 8, lane), al = *wfrag(w, 0, ks, 1, 8, lane);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const unsigned char* row = xt + (ct * 16 + c16) * HT_ROWB + (ks * 32 + g * 8) * 2;
        const bf16x8 xh = *reinterpret_cast<const bf16x8*>(row);
        const bf16x8 xl = *reinterpret_cast<const bf16x8*>(row + HT_PLANE);
        oacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh, oacc[ct], 0, 0, 0);
        oacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl, oacc[ct], 0, 0, 0);
        oacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh, oacc[ct], 0, 0, 0);
      }
    }
  }
  __syncthreads();  // tile no longer needed: reuse LDS for the 4 partial sums [wave][n 16][px 64]
  float* red = reinterpret_cast<float*>(xt);
  const int n_out = p.n_out[head], act = p.act[head];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = 4 * g + r;
      if (n < n_out) red[(wave * 16 + n) * HT_PX + ct * 16 + c16] = oacc[ct][r];
    }
  __syncthreads();
  const float* bo = p.b_out[head];
  float* out = p.out[head];
  float* out2 = p.out2[head];
  const int px = tid & 63;
  int b, pix;
  if (pm(px, b, pix)) {
    for (int n = tid >> 6; n < n_out; n += 4) {
      const float raw = red[n * HT_PX + px] + red[(16 + n) * HT_PX + px] + red[(32 + n) * HT_PX + px] +
                        red[(48 + n) * HT_PX + px] + bo[n];
      const size_t o = ((size_t)b * n_out + n) * p.HW + pix;
      float v = raw;
      if (act == CF_ACT_RELU) v = fmaxf(raw, 0.0f);
      else if (act == CF_ACT_SIGMOID_CLAMP) v = fminf(fmaxf(cf_sigmoid(raw), 1e-4f), 1.0f - 1e-4f);
      out[o] = v;
      if (act == CF_ACT_RAW_AND_SIGDEPTH) out2[o] = 1.0f / (cf_sigmoid(raw) + 1e-6f) - 1.0f;
    }
  }
}

#ifdef CF_LEGACY_HEADS   // (32x32x16 / slot-table head kernels of rounds 1-3: nothing dispatches them in the default build)
// Tail only: the 256-channel hidden tile comes from a split-bf16 tensor in HBM.
__global__ __launch_bounds__(256) void head_tail_kernel(HeadTailK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char xt[];  // [2 planes][64 px][528 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int tiles = (p.M + HT_PX - 1) / HT_PX;
  const int head = blockIdx.x / tiles, tile = blockIdx.x - head * tiles;
  const int m0 = tile * HT_PX;
  const int cb = p.c_base[head];

  if (p.n_hidden > 0) {
    // all 16 requests of a thread are issued before the first LDS write: the whole 64 KiB tile is
    // in flight at once
    u32x4 v[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int i = tid + it * 256;
      const int unit = i & 31, plane = (i >> 5) & 1, px = i >> 6;
      const int m = m0 + px;
      v[it] = u32x4{0u, 0u, 0u, 0u};
      if (m < p.M)
        v[it] = *reinterpret_cast<const u32x4*>(p.x + (((size_t)m * 2 + plane) * p.x_stride + cb) * 2 + unit * 16);
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int i = tid + it * 256;
      const int unit = i & 31, plane = (i >> 5) & 1, px = i >> 6;
      *reinterpret_cast<u32x4*>(xt + plane * HT_PLANE + px * HT_ROWB + unit * 16) = v[it];
    }
    __syncthreads();
    head_tail_from_lds(p, xt, head, FlatMap{m0, p.M, p.HW});
    return;
  }
  // no hidden layer: the B fragments (pixels x this wave's 64 input channels) come straight from
  // HBM - every byte of the tile is read exactly once by exactly one lane, no LDS staging
  f32x16 oacc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[b][r] = 0.0f;
  const unsigned char* w = p.w_out[head];
  bf16x8 xh[2][4], xl[2][4];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int m = m0 + ct * 32 + li;
    const unsigned char* row = p.x + ((size_t)(m < p.M ? m : 0) * 2 * p.x_stride + cb) * 2;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = (wave * 4 + s) * 16 + h * 8;
      xh[ct][s] = *reinterpret_cast<const bf16x8*>(row + k * 2);
      xl[ct][s] = *reinterpret_cast<const bf16x8*>(row + ((size_t)p.x_stride + k) * 2);
    }
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int ks = wave * 4 + s;
    const bf16x8 ah = *wfrag(w, 0, ks, 0, 16, lane), al = *wfrag(w, 0, ks, 1, 16, lane);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh[ct][s], oacc[ct], 0, 0, 0);
      oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl[ct][s], oacc[ct], 0, 0, 0);
      oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh[ct][s], oacc[ct], 0, 0, 0);
    }
  }
  reduce_and_store(p, xt, oacc, head, FlatMap{m0, p.M, p.HW});
}
#endif  // CF_LEGACY_HEADS

// ---------------------------------------------------------------------------------------------
// Whole head in one launch: 3x3 conv (64 [+3] -> 256) + ReLU, then the tail above.  The first layer
// is the same swapped GEMM: weights as pre-packed A fragments straight from L2 (4 k-steps ahead in
// registers), the pixel operand (implicit im2col of the split-bf16 feature map, 8-channel slots) is
// staged per 32-deep chunk through a small double-buffered LDS tile [64 px][32 k] shared by the 4
// waves - one barrier per chunk.  The 256-channel hidden map is born in LDS and never reaches HBM.
// ---------------------------------------------------------------------------------------------
struct HeadFusedK {
  HeadTailK t;
  const unsigned char* src[2];      // split-bf16 NHWC sources of the 3x3 layer (feat, pc_hm)
  int src_c[2];
  const cf_slot* slots;             // 8-channel slots, 4 per chunk, n_chunks even
  int n_chunks, H, W;
  const unsigned char* w_first[CF_MAX_HEADS];  // fragment-packed [8 rt][K_pad/16 ks]
  const float* b_first[CF_MAX_HEADS];
};

constexpr int HF_ROWB = 80;                       // 32 bf16 + 16 B pad per pixel row per plane
constexpr int HF_PLANE = HT_PX * HF_ROWB;         // 5120
constexpr int HF_BUF = 2 * HF_PLANE;              // 10240 per chunk buffer
constexpr int HF_MAX_CHUNKS = 64;
constexpr int HF_LDS = HT_LDS + HF_MAX_CHUNKS * 4 * (int)sizeof(cf_slot);   // tile area + slot table

#ifdef CF_LEGACY_HEADS   // (the slot-table head kernel)
__global__ __launch_bounds__(256) void head_fused_kernel(HeadFusedK q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char xt[];  // B chunk buffers, later the hidden tile
  const HeadTailK& p = q.t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int tiles = (p.M + HT_PX - 1) / HT_PX;
  const int head = blockIdx.x / tiles, tile = blockIdx.x - head * tiles;
  const int m0 = tile * HT_PX;
  const unsigned char* w1 = q.w_first[head];
  const int n_ks = q.n_chunks * 2;

  // staging role of this thread: one pixel row, one 8-channel unit, both planes
  const int spx = tid >> 2, su = tid & 3;
  int y0, x0, boff;
  {
    const int m = m0 + spx;
    if (m < p.M) {
      const int b = m / p.HW, rem = m - b * p.HW;
      y0 = rem / q.W;
      x0 = rem - y0 * q.W;
      boff = b * p.HW;
    } else {
      y0 = -(1 << 28);
      x0 = 0;
      boff = 0;
    }
  }
  // slot table -> LDS (behind the tile area): a per-chunk global read of it would put a dependent
  // L2 round trip (and a vmcnt(0) drain of the weight prefetches) in front of every staging load
  cf_slot* lds_slots = reinterpret_cast<cf_slot*>(xt + HT_LDS);
  for (int i = tid; i < q.n_chunks * 4; i += 256) lds_slots[i] = q.slots[i];
  __syncthreads();
  u32x4 sh, sl;
  auto load_b = [&](int c) {
    const cf_slot s = lds_slots[c * 4 + su];
    const int src = __builtin_amdgcn_readfirstlane(lds_slots[c * 4].src);
    const unsigned char* sp = src == 1 ? q.src[1] : q.src[0];
    const int sc = src == 1 ? q.src_c[1] : q.src_c[0];
    const int y = y0 + s.dy, x = x0 + s.dx;
    const bool ok = (s.c_off >= 0) && ((unsigned)y < (unsigned)q.H) && ((unsigned)x < (unsigned)q.W);
    sh = u32x4{0u, 0u, 0u, 0u};
    sl = sh;
    if (ok) {
      const unsigned char* a = sp + ((size_t)(boff + y * q.W + x) * (2 * sc) + s.c_off) * 2;
      sh = *reinterpret_cast<const u32x4*>(a);
      sl = *reinterpret_cast<const u32x4*>(a + (size_t)sc * 2);
    }
  };
  auto store_b = [&](unsigned char* buf) {
    *reinterpret_cast<u32x4*>(buf + spx * HF_ROWB + su * 16) = sh;
    *reinterpret_cast<u32x4*>(buf + HF_PLANE + spx * HF_ROWB + su * 16) = sl;
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  // weight fragments, 4 k-steps (= 2 chunks) ahead: set t holds k-step (4n + t)
  bf16x8 wh[4][2], wl[4][2];
  auto load_w = [&](bf16x8 (&dh)[2], bf16x8 (&dl)[2], int ks) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      dh[rt] = *wfrag(w1, wave * 2 + rt, ks, 0, n_ks, lane);
      dl[rt] = *wfrag(w1, wave * 2 + rt, ks, 1, n_ks, lane);
    }
  };
  auto mma_kstep = [&](const unsigned char* buf, int s, const bf16x8 (&ah)[2], const bf16x8 (&al)[2]) {
    bf16x8 xh[2], xl[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const unsigned char* row = buf + (ct * 32 + li) * HF_ROWB + s * 32 + h * 16;
      xh[ct] = *reinterpret_cast<const bf16x8*>(row);
      xl[ct] = *reinterpret_cast<const bf16x8*>(row + HF_PLANE);
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[rt], xh[ct], acc[rt][ct], 0, 0, 0);
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], xl[ct], acc[rt][ct], 0, 0, 0);
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], xh[ct], acc[rt][ct], 0, 0, 0);
      }
  };

  load_b(0);
  load_w(wh[0], wl[0], 0);
  load_w(wh[1], wl[1], 1);
  load_w(wh[2], wl[2], 2);
  load_w(wh[3], wl[3], 3);
  store_b(xt);
  load_b(1);
  __syncthreads();
  // two chunks (four k-steps) per iteration so the four fragment sets are addressed statically.
  // Straight-line body (look-ahead indices clamped, not branched) with a sched_barrier after every
  // k-step: without it the compiler sinks the prefetch loads down to their first use.
  const int last_c = q.n_chunks - 1;
  for (int c = 0; c < q.n_chunks; c += 2) {
    unsigned char* b0 = xt;
    unsigned char* b1 = xt + HF_BUF;
    const int ks = c * 2;
    // chunk c  (buffer 0)
    mma_kstep(b0, 0, wh[0], wl[0]);
    load_w(wh[0], wl[0], min(ks + 4, n_ks - 1));
    __builtin_amdgcn_sched_barrier(0);
    mma_kstep(b0, 1, wh[1], wl[1]);
    load_w(wh[1], wl[1], min(ks + 5, n_ks - 1));
    store_b(b1);                       // chunk c+1 (requested a chunk ago)
    load_b(min(c + 2, last_c));
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    // chunk c+1 (buffer 1)
    mma_kstep(b1, 0, wh[2], wl[2]);
    load_w(wh[2], wl[2], min(ks + 6, n_ks - 1));
    __builtin_amdgcn_sched_barrier(0);
    mma_kstep(b1, 1, wh[3], wl[3]);
    load_w(wh[3], wl[3], min(ks + 7, n_ks - 1));
    store_b(b0);                       // chunk c+2 (a repeat of the last chunk at the end: never read)
    load_b(min(c + 3, last_c));
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
  // hidden = ReLU(acc + b) -> LDS tile (the staging buffers are dead: last barrier above)
  store_hidden_tile(xt, acc, q.b_first[head], wave, li, h);
  __syncthreads();
  head_tail_from_lds(p, xt, head, FlatMap{m0, p.M, p.HW});
}


#endif  // CF_LEGACY_HEADS

// ---------------------------------------------------------------------------------------------
// Whole head with n_hidden == 0 on a 2-D PATCH: one workgroup owns an 8 x 16 pixel tile (128 px) of
// one image and one head.  The (8+2) x (16+2) input patch - all 64 feature channels (and the 8-channel
// pc_hm plane pair) in split-bf16 - is copied to LDS once, zero-filled outside the image, and all 9
// taps x 4 slices read their B fragments from it at compile-time offsets: no slot table, no barrier
// and no staging inside the K loop.  A wave owns 64 hidden channels x 128 pixels, so the first-layer
// weights (the dominant L2 stream of the 64-pixel kernel above) are fetched once per 128 pixels.
// The 256 -> n_out layer runs straight from the accumulator registers: ReLU(acc + b) split to bf16 IS
// a B fragment if the output weights are packed with the matching k permutation (w_out_perm: position
// 8h + j of a 16-group holds channel 4h + (j & 3) + 8(j >> 2)); the four waves' partial sums meet in LDS.
// ---------------------------------------------------------------------------------------------
constexpr int HP_TH = 8, HP_TW = 16, HP_PW = HP_TW + 2, HP_ROWS = (HP_TH + 2) * HP_PW;   // 180 patch rows
constexpr int HP_PX = HP_TH * HP_TW;                                                    // 128
constexpr int HP_RED = 4 * 32 * HP_PX * 4;                                              // 64 KiB of partial sums
constexpr int HP_LDS = HT_LDS > HP_RED ? HT_LDS : HP_RED;   // patch (<= 55 KiB) / partial sums / 64-pixel hidden tile
constexpr int HP16_RED = 4 * 16 * HP_PX * 4;   // 16x16x32 kernel: partial sums [wave][16][128] BEHIND the patch
constexpr int hp16_patch_bytes(bool pc) { return HP_ROWS * (4 * 64 + (pc ? 32 : 0) + 16); }   // 48,960 / 54,720
// without pc_hm: 48,960 + 32,768 = 81,728 B <= half of the CU's 160 KiB: still two workgroups per CU
constexpr int hp16_lds(bool pc, bool hidden) {
  return hidden ? (HT_LDS > hp16_patch_bytes(pc) ? HT_LDS : hp16_patch_bytes(pc)) : hp16_patch_bytes(pc) + HP16_RED;
}

struct HeadPatchK {
  HeadTailK t;
  const unsigned char* src[2];
  int src_c[2];
  int H, W, tiles_x, tiles_y, n_ks;
  int group;                                 // 0: head-major grid; g > 0: tile-major within groups of g heads (32x32x16 kernel)
  int hloop;                                 // 16x16x32 kernel: consecutive heads one workgroup walks on its patch
  const unsigned char* w_first[CF_MAX_HEADS];
  const float* b_first[CF_MAX_HEADS];
  const unsigned char* w_out_perm[CF_MAX_HEADS];
  float first_scale[CF_MAX_HEADS];           // MX kernel: 2^-(s+4) of head i's first layer
};

#ifdef CF_LEGACY_HEADS   // (32x32x16 / slot-table head kernels of rounds 1-3: nothing dispatches them in the default build)
template <int NS, bool PC>
__global__ __launch_bounds__(256, 2) void head_patch_kernel(HeadPatchK q) {
  constexpr int ROWB = NS * 64 + (PC ? 32 : 0) + 16;     // odd multiple of 16 B
  constexpr int NK = 9 * NS + (PC ? 5 : 0);              // k-steps of the first layer
  extern __shared__ __attribute__((aligned(16))) unsigned char xt[];
  const HeadTailK& p = q.t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int per_img = q.tiles_x * q.tiles_y;
  const int per_head = per_img * (p.M / p.HW);
  int head, rem;
  if (q.group == 0) {                        // head-major: every tile of head 0, then head 1, ...
    head = blockIdx.x / per_head;
    rem = blockIdx.x - head * per_head;
  } else {
    // tile-major inside groups of `group` heads, XCD-aware: the hardware deals workgroups round-robin over the 8
    // XCDs, so the blocks one XCD sees (b, b + 8, ...) are made to walk (tile, head of the group) with the head
    // fastest - the heads of a tile then run back to back on ONE XCD and all but the first find the patch in L2
    const int g = q.group;
    const int n_groups = (p.n_heads + g - 1) / g;
    const int per_group = per_head * g;      // (the last group may be short: its surplus blocks exit)
    const int grp = blockIdx.x / per_group;
    const int lb = cf_xcd_remap(blockIdx.x - grp * per_group, per_group);
    (void)n_groups;
    rem = lb / g;
    head = grp * g + (lb - rem * g);
    if (head >= p.n_heads) return;
  }
  const int b = rem / per_img;
  rem -= b * per_img;
  const int y0 = (rem / q.tiles_x) * HP_TH, x0 = (rem % q.tiles_x) * HP_TW;
  const unsigned char* w1 = q.w_first[head];

  // ---- patch -> LDS (one pass, every load in flight before the first LDS write)
  {
    constexpr int UPR = NS * 4;                            // 16-byte units per row: hi plane then lo plane
    constexpr int NIT = (HP_ROWS * UPR + 255) / 256;
    u32x4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * 256;
      const int row = idx / UPR, u = idx % UPR;
      const int y = y0 - 1 + row / HP_PW, x = x0 - 1 + row % HP_PW;
      v[it] = u32x4{0u, 0u, 0u, 0u};
      if (row < HP_ROWS && (unsigned)y < (unsigned)q.H && (unsigned)x < (unsigned)q.W)
        v[it] = *reinterpret_cast<const u32x4*>(q.src[0] + ((size_t)(b * p.HW + y * q.W + x) * 2 * q.src_c[0]) * 2 +
                                                (u / (NS * 2)) * q.src_c[0] * 2 + (u % (NS * 2)) * 16);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * 256;
      const int row = idx / UPR, u = idx % UPR;
      const int plane = u / (NS * 2), uu = u % (NS * 2);
      if (row < HP_ROWS) *reinterpret_cast<u32x4*>(xt + row * ROWB + (uu >> 1) * 64 + plane * 32 + (uu & 1) * 16) = v[it];
    }
    if (PC) {
      for (int idx = tid; idx < HP_ROWS * 2; idx += 256) {
        const int row = idx >> 1, plane = idx & 1;
        const int y = y0 - 1 + row / HP_PW, x = x0 - 1 + row % HP_PW;
        u32x4 w = {0u, 0u, 0u, 0u};
        if ((unsigned)y < (unsigned)q.H && (unsigned)x < (unsigned)q.W)
          w = *reinterpret_cast<const u32x4*>(q.src[1] + ((size_t)(b * p.HW + y * q.W + x) * 2 + plane) * q.src_c[1] * 2);
        *reinterpret_cast<u32x4*>(xt + row * ROWB + NS * 64 + plane * 16) = w;
      }
    }
  }

  int rowb[4];                               // LDS byte offset of this lane's pixel row (tap (-1,-1)) + k half
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int pl = ct * 32 + li;
    rowb[ct] = ((pl >> 4) * HP_PW + (pl & 15)) * ROWB + h * 16;
  }
  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.0f;

  bf16x8 wh[3][2], wl[3][2];                 // weight fragments three k-steps ahead: set ks % 3
  auto load_w = [&](bf16x8 (&dh)[2], bf16x8 (&dl)[2], int ks) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      dh[rt] = *wfrag(w1, wave * 2 + rt, ks, 0, q.n_ks, lane);
      dl[rt] = *wfrag(w1, wave * 2 + rt, ks, 1, q.n_ks, lane);
    }
  };
#pragma unroll
  for (int t = 0; t < 3; ++t) load_w(wh[t], wl[t], t);
  __syncthreads();

#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    bf16x8 xh[4], xl[4];
    if (ks < 9 * NS) {
      constexpr int dummy = 0;
      (void)dummy;
      const int tap = ks / NS, sl = ks % NS;
      const int off = ((tap / 3) * HP_PW + tap % 3) * ROWB + sl * 64;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        xh[ct] = *reinterpret_cast<const bf16x8*>(xt + rowb[ct] + off);
        xl[ct] = *reinterpret_cast<const bf16x8*>(xt + rowb[ct] + off + 32);
      }
    } else {                                 // pc_hm: k half h of step i is tap 2i + h (8 channels each)
      const int i = ks - 9 * NS;
      const int t0 = 2 * i, t1 = 2 * i + 1 < 9 ? 2 * i + 1 : 8;
      const int o0 = ((t0 / 3) * HP_PW + t0 % 3) * ROWB, o1 = ((t1 / 3) * HP_PW + t1 % 3) * ROWB;
      const int off = (h ? o1 - 16 : o0) + NS * 64;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        xh[ct] = *reinterpret_cast<const bf16x8*>(xt + rowb[ct] + off);
        xl[ct] = *reinterpret_cast<const bf16x8*>(xt + rowb[ct] + off + 16);
      }
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[ks % 3][rt], xh[ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[ks % 3][rt], xl[ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[ks % 3][rt], xh[ct], acc[rt][ct], 0, 0, 0);
    if (ks + 3 < NK) load_w(wh[ks % 3], wl[ks % 3], ks + 3);
    __builtin_amdgcn_sched_barrier(0);
  }

  if (p.n_hidden > 0) {
    // hidden layers need all 256 channels of a pixel: the two 64-pixel halves of the tile go through
    // the LDS-resident chain of cf_head_tail one after the other (LDS stays at 66 KiB: 2 workgroups/CU)
    __syncthreads();                         // every wave is done with the patch
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x16 a2[2][2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) a2[rt][ct] = acc[rt][2 * half + ct];
      store_hidden_tile(xt, a2, q.b_first[head], wave, li, h);
      __syncthreads();
      head_tail_from_lds(p, xt, head, TileMap{b, y0 + 4 * half, x0, q.H, q.W});
      __syncthreads();
    }
    return;
  }

  // ---- output layer from registers: this wave's 64 hidden channels = 4 k-steps of 16
  f32x16 oacc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[c][r] = 0.0f;
  {
    const float* b1 = q.b_first[head] + wave * 64 + 4 * h;
    const unsigned char* wo = q.w_out_perm[head];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int ks2 = wave * 4 + rt * 2 + jj;
        const bf16x8 ah = *wfrag(wo, 0, ks2, 0, 16, lane), al = *wfrag(wo, 0, ks2, 1, 16, lane);
        const f32x4 ba = *reinterpret_cast<const f32x4*>(b1 + rt * 32 + 16 * jj);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b1 + rt * 32 + 16 * jj + 8);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          float v[8], hi[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            v[j] = fmaxf(acc[rt][ct][8 * jj + j] + (j < 4 ? ba[j] : bb[j - 4]), 0.0f);
            hi[j] = bf16_rne(v[j]);
          }
          const u32x4 ph = {pack2(hi[0], hi[1]), pack2(hi[2], hi[3]), pack2(hi[4], hi[5]), pack2(hi[6], hi[7])};
          const u32x4 pl = {pack2(v[0] - hi[0], v[1] - hi[1]), pack2(v[2] - hi[2], v[3] - hi[3]),
                            pack2(v[4] - hi[4], v[5] - hi[5]), pack2(v[6] - hi[6], v[7] - hi[7])};
          const bf16x8 xh = __builtin_bit_cast(bf16x8, ph), xl = __builtin_bit_cast(bf16x8, pl);
          oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh, oacc[ct], 0, 0, 0);
          oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl, oacc[ct], 0, 0, 0);
          oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh, oacc[ct], 0, 0, 0);
        }
      }
  }
  __syncthreads();                           // every wave is done with the patch: reuse LDS for the partial sums
  const int n_out = p.n_out[head], act = p.act[head];
  float* red = reinterpret_cast<float*>(xt); // [wave][n 32][px 128]
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = (r & 3) + 8 * (r >> 2) + 4 * h;
      if (n < n_out) red[(wave * 32 + n) * HP_PX + ct * 32 + li] = oacc[ct][r];
    }
  __syncthreads();
  {
    const int px = tid & (HP_PX - 1);
    const int y = y0 + (px >> 4), x = x0 + (px & 15);
    if (y < q.H && x < q.W) {
      const float* bo = p.b_out[head];
      float* out = p.out[head];
      float* out2 = p.out2[head];
      for (int n = tid >> 7; n < n_out; n += 2) {
        const float raw = red[n * HP_PX + px] + red[(32 + n) * HP_PX + px] + red[(64 + n) * HP_PX + px] +
                          red[(96 + n) * HP_PX + px] + bo[n];
        const size_t o = ((size_t)b * n_out + n) * p.HW + (size_t)y * q.W + x;
        float v = raw;
        if (act == CF_ACT_RELU) v = fmaxf(raw, 0.0f);
        else if (act == CF_ACT_SIGMOID_CLAMP) v = fminf(fmaxf(cf_sigmoid(raw), 1e-4f), 1.0f - 1e-4f);
        out[o] = v;
        if (act == CF_ACT_RAW_AND_SIGDEPTH) out2[o] = 1.0f / (cf_sigmoid(raw) + 1e-6f) - 1.0f;
      }
    }
  }
}


#endif  // CF_LEGACY_HEADS

// ---------------------------------------------------------------------------------------------
// The same whole-head patch kernel on v_mfma_f32_16x16x32_bf16.  Under the chip's power management a dense MFMA
// loop on random data holds a higher clock with the 16x16x32 shape than with 32x32x16 at equal cycles per FLOP:
// measured on this part 1.72 vs 1.51 PFLOP/s with every operand re-read from LDS (tools/micro/mfma_shape.hip,
// MI355X_MICROARCH.md "DVFS give-back" item 7) - and this kernel is bound by exactly that loop.  Same tile (8 x 16
// pixels), same LDS patch, same operand bytes per FLOP: a wave owns 64 hidden channels x 128 pixels as 4 x 8
// accumulators of 16 x 16; a k-step is 32 deep = half the channels of one tap (or four taps of the 8-channel pc_hm
// plane pair).  Weights come as [16-row tile][k32 step][hi|lo][lane][8] fragments (packing.pack_fragments16), one
// k-step ahead in registers.  The 256 -> n_out layer again runs from the accumulator registers: two stacked 16 x 16
// tiles give a lane 4 + 4 channels of one pixel = one B fragment, with w_out packed in that k order
// (pack_fragments16(acc_order=True)); n_out <= 16.
// ---------------------------------------------------------------------------------------------
//
// MX = true: the FIRST layer on "fp16 main term + block-scaled FP6 cross terms" (1.5 MFMA passes per product instead of the 3
// of bf16x3; numerics and the gate that confines the scheme to the first layer: packing.pack_head_first_mx).  The
// feature source is the 272-byte-per-pixel image cf_pack_feat_mx writes - four 64-byte segments g = 0..3, each [8 fp16: channels
// 8g..8g+7 of 16 x][8 fp16: channels 32+8g..][32 FP6 e2m3 fields (24 B) + 8 B pad: block g of q6(xl) channels 0-31, 32-63,
// q6(xh) channels 0-31, 32-63], then one E8M0 scale byte per block and padding - which IS the LDS patch row: lane group g of
// every B fragment reads inside segment g (bank-conflict-free ds_read_b128, see colperm below).  Per tap: 2 k-steps of v_mfma_f32_16x16x32_f16 (weights' fp16 hi) and ONE v_mfma_scale_f32_16x16x128_f8f6f4
// whose four 32-deep K blocks are q6(Wh) . q6(xl) (two 32-channel halves) and q6(Wl) . q6(xh): lane g = l >> 4 of either operand
// holds K block g, with the block's scale byte in its lane.  Weights stream from L2 in (wave, tap) slabs, two items (of
// main / main / cross) ahead.  pc_hm stays on bf16x3 with its weights pre-multiplied by 2^(s+4); the accumulators are scaled by
// first_scale = 2^-(s+4) where the bias is added.  Hidden and output layers: unchanged bf16x3.
// HID: -1 = n_hidden decided at run time (the bf16x3 instantiations); 0 / 1 = compiled for heads without / with hidden layers
// (the MX instantiations: the register allocator then sees one of the two epilogues, not both).
template <int NS, bool PC, bool TP, bool MX = false, int HID = -1>
__global__ __launch_bounds__(256, 2) void head_patch16_kernel(HeadPatchK q) {
  static_assert(NS == 4, "64 feature channels");
  constexpr int MX_SLAB = 14592, PC_OFF = MX ? 272 : NS * 64;      // (wave, tap) weight slab bytes; pc_hm planes inside a patch row
  // TP: the 128-pixel tile stands upright (16 rows x 8 columns) instead of 8 x 16 - same patch size (18 x 10 rows), chosen by
  // the host when it covers the map with fewer tiles (112 x 200: 7 x 25 = 175 exact tiles instead of 14 x 13 = 182)
  constexpr int TSH = TP ? 3 : 4, TMASK = (1 << TSH) - 1;          // pixel index -> (row = px >> TSH, column = px & TMASK)
  constexpr int T_H = TP ? 16 : 8, T_W = TP ? 8 : 16, P_W = T_W + 2;
  static_assert((T_H + 2) * P_W == HP_ROWS, "both tile shapes have the same patch size");
  constexpr int ROWB = NS * 64 + (PC ? 32 : 0) + 16;     // odd multiple of 16 B
  constexpr int NKF = 9 * NS / 2;                        // k32-steps of the feature channels
  constexpr int NK = NKF + (PC ? 3 : 0);
  extern __shared__ __attribute__((aligned(16))) unsigned char xt[];
  const HeadTailK& p = q.t;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: scalar weight tile addressing)
  const int g = lane >> 4;
  // MX, flat tile: lane l & 15 holds pixel COLUMN colperm(l & 15) of the 16-pixel tile row, not column l & 15.  A ds_read_b128 is
  // served in groups of 16 lanes made of 8 lanes of one K group g and 8 of g + 1 ({0-3, 12-15, 20-27}, ...); with the K groups
  // of a patch row 64 B apart and the row pitch = 16 B (mod 256 B), the 16 lanes of a group hit 16 different 16-byte bank
  // quads iff the columns of lanes {4..11} form a set invariant under +4: {0,4,8,12,1,5,9,13} (the others get the rest).
  const int c16 = (MX && !TP) ? ((lane & 12) == 4 || (lane & 12) == 8 ? (((lane & 15) - 4) & 3) * 4 + (((lane & 15) - 4) >> 2)
                                                                      : (lane & 3) * 4 + 2 + ((lane & 15) >> 3))
                              : (lane & 15);
  const int per_img = q.tiles_x * q.tiles_y;
  const int per_head = per_img * (p.M / p.HW);
  // grid = (head range, tile): a workgroup keeps its patch in LDS and walks q.hloop consecutive heads on it (heads
  // without hidden layers only: the hidden chain rewrites the patch area).  Head-range-major order, so the
  // workgroups in flight work on the same few heads and their first-layer weights stay hot in L2, while the patch
  // is read from HBM once per range instead of once per head.
  const int hg = blockIdx.x / per_head;
  // within a head range consecutive tiles run on ONE XCD: neighbouring patches share their halo rows in that XCD's L2
  int rem = cf_xcd_remap(blockIdx.x - hg * per_head, per_head);
  const int head0 = hg * q.hloop, head1 = min(head0 + q.hloop, p.n_heads);
  const int b = rem / per_img;
  rem -= b * per_img;
  const int y0 = (rem / q.tiles_x) * T_H, x0 = (rem % q.tiles_x) * T_W;

  // ---- patch -> LDS (one pass, every load in flight before the first LDS write)
  if constexpr (MX) {
    constexpr int UPR = 17;                                // 16-byte units of a 272-byte mx row: the LDS row image itself
    constexpr int NIT = (HP_ROWS * UPR + 255) / 256;
    u32x4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * 256;
      const int row = idx / UPR, u = idx % UPR;
      const int y = y0 - 1 + row / P_W, x = x0 - 1 + row % P_W;
      v[it] = u32x4{0u, 0u, 0u, 0u};                       // outside the image: zero fields with scale byte 0 (2^-127): exact zeros
      if (row < HP_ROWS && (unsigned)y < (unsigned)q.H && (unsigned)x < (unsigned)q.W)
        v[it] = *reinterpret_cast<const u32x4*>(q.src[0] + (size_t)(b * p.HW + y * q.W + x) * 272 + u * 16);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * 256;
      const int row = idx / UPR, u = idx % UPR;
      if (row < HP_ROWS) *reinterpret_cast<u32x4*>(xt + row * ROWB + u * 16) = v[it];
    }
  } else {
    constexpr int UPR = NS * 4;                            // 16-byte units per row: hi plane then lo plane
    constexpr int NIT = (HP_ROWS * UPR + 255) / 256;
    u32x4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * 256;
      const int row = idx / UPR, u = idx % UPR;
      const int y = y0 - 1 + row / P_W, x = x0 - 1 + row % P_W;
      v[it] = u32x4{0u, 0u, 0u, 0u};
      if (row < HP_ROWS && (unsigned)y < (unsigned)q.H && (unsigned)x < (unsigned)q.W)
        v[it] = *reinterpret_cast<const u32x4*>(q.src[0] + ((size_t)(b * p.HW + y * q.W + x) * 2 * q.src_c[0]) * 2 +
                                                (u / (NS * 2)) * q.src_c[0] * 2 + (u % (NS * 2)) * 16);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * 256;
      const int row = idx / UPR, u = idx % UPR;
      const int plane = u / (NS * 2), uu = u % (NS * 2);
      if (row < HP_ROWS) *reinterpret_cast<u32x4*>(xt + row * ROWB + (uu >> 1) * 64 + plane * 32 + (uu & 1) * 16) = v[it];
    }
  }
  if (PC) {
    for (int idx = tid; idx < HP_ROWS * 2; idx += 256) {
      const int row = idx >> 1, plane = idx & 1;
      const int y = y0 - 1 + row / P_W, x = x0 - 1 + row % P_W;
      u32x4 w = {0u, 0u, 0u, 0u};
      if ((unsigned)y < (unsigned)q.H && (unsigned)x < (unsigned)q.W)
        w = *reinterpret_cast<const u32x4*>(q.src[1] + ((size_t)(b * p.HW + y * q.W + x) * 2 + plane) * q.src_c[1] * 2);
      *reinterpret_cast<u32x4*>(xt + row * ROWB + PC_OFF + plane * 16) = w;
    }
  }

  // B fragment of a 16x16x32 MFMA: lane (g = l >> 4, c = l & 15) holds 8 consecutive channels 8g .. 8g+7 of the
  // k-step's 32 for pixel column c of the 16-pixel tile row
  int rowb[8];                               // LDS byte offset of this lane's pixel in tile row ct (tap (-1,-1))
#pragma unroll
  for (int ct = 0; ct < 8; ++ct)          // pixel ct * 16 + c16: a compile-time stride per ct (immediate offsets of the ds_reads)
    rowb[ct] = ((c16 >> TSH) * P_W + (c16 & TMASK)) * ROWB + ct * ((16 >> TSH) * P_W * ROWB);
  const int koff = (g >> 1) * 64 + (g & 1) * 16;         // 8-channel group inside a 32-channel half (hi plane; lo at +32)
  int pc_off[3];                             // pc_hm steps: k-group g of step i is tap 4i + g (taps 9..11: zero weights)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int t = min(4 * i + g, 8);
    pc_off[i] = ((t / 3) * P_W + t % 3) * ROWB + PC_OFF;
  }
  __syncthreads();                           // the patch is complete
  auto first_layer = [&](int head, f32x4 (&acc)[4][8]) __attribute__((always_inline)) {
  if constexpr (MX) {
    const unsigned char* wb = q.w_first[head] + (size_t)wave * (9 * MX_SLAB);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][c][r] = 0.0f;
    hf16x8 am[2][4];                         // fp16 hi fragments of the tap's two k-steps
    u32x4 ax0[4];                            // FP6 fragments of the tap's cross term: 24 B per lane and row tile
    u32x2 ax1[4];
    int sa;                                  // their E8M0 scale bytes, byte rt
    // buffer loads: a scalar resource per (wave, tap) slab + a per-lane 32-bit offset + small constants - no 64-bit address
    // arithmetic and no address register pairs in a loop that has every register in use
    auto slab = [&](int tap) {
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(wb + tap * MX_SLAB), 0, MX_SLAB, 0x00020000);
    };
    const int l16 = lane * 16, l8 = lane * 8, l4 = lane * 4;
    auto ldm = [&](hf16x8 (&d)[4], int tap, int ks2) {
      const __amdgpu_buffer_rsrc_t rs = slab(tap);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
        d[rt] = __builtin_bit_cast(hf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, l16, (rt * 2 + ks2) * 1024, 0));
    };
    auto ldx = [&](int tap) {
      const __amdgpu_buffer_rsrc_t rs = slab(tap);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        ax0[rt] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, l16, 8192 + rt * 1536, 0));
        ax1[rt] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, l8, 8192 + rt * 1536 + 1024, 0));
      }
      sa = (int)__builtin_amdgcn_raw_buffer_load_b32(rs, l4, 14336, 0);
    };
    // one half (tile rows 4 hf .. 4 hf + 3) of a main k-step / one pair (tile rows 2 pr, 2 pr + 1) of a cross step
#ifdef CF_MX_ARM_NOB       // dev timing arm (garbage results): the B fragments are read once, before the loop
    hf16x8 xb_fix[4];
    i32x8 xc_fix[2];
    int sc_fix[2];
    for (int ct = 0; ct < 4; ++ct) xb_fix[ct] = *reinterpret_cast<const hf16x8*>(xt + rowb[ct] + g * 64);
    for (int c2 = 0; c2 < 2; ++c2) {
      const unsigned char* r = xt + rowb[c2];
      const u32x4 b0 = *reinterpret_cast<const u32x4*>(r + 64 * g + 32);
      const u32x2 b1 = *reinterpret_cast<const u32x2*>(r + 64 * g + 48);
      xc_fix[c2] = i32x8{(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], 0, 0};
      sc_fix[c2] = (int)(*reinterpret_cast<const unsigned*>(r + 256) >> (8 * g));
    }
#endif
    auto main_half = [&](const hf16x8 (&A)[4], int off, int hf) {
      hf16x8 xb[4];
#pragma unroll
#ifdef CF_MX_ARM_NOB
      for (int ct = 0; ct < 4; ++ct) xb[ct] = xb_fix[(ct + hf + off / 16) & 3];      // (rotating: the operands still change from MFMA to MFMA)
#else
      for (int ct = 0; ct < 4; ++ct) xb[ct] = *reinterpret_cast<const hf16x8*>(xt + rowb[4 * hf + ct] + off);
#endif
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc[rt][4 * hf + ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[rt], xb[ct], acc[rt][4 * hf + ct], 0, 0, 0);
    };
    auto cross_pair = [&](int toff, int pr) {
      i32x8 xb[2];
      int sb[2];
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const unsigned char* r = xt + rowb[2 * pr + c2] + toff;
        const u32x4 b0 = *reinterpret_cast<const u32x4*>(r + 64 * g + 32);
        const u32x2 b1 = *reinterpret_cast<const u32x2*>(r + 64 * g + 48);
        xb[c2] = i32x8{(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], 0, 0};
        sb[c2] = (int)(*reinterpret_cast<const unsigned*>(r + 256) >> (8 * g));     // this lane's block: byte g -> byte 0
#ifdef CF_MX_ARM_NOB
        xb[c2] = xc_fix[(c2 + pr) & 1];
        sb[c2] = sc_fix[(c2 + pr) & 1];
#endif
      }
#define CF_MX_ROW(RT)                                                                                                   \
      {                                                                                                                 \
        const i32x8 a6 = {(int)ax0[RT][0], (int)ax0[RT][1], (int)ax0[RT][2], (int)ax0[RT][3], (int)ax1[RT][0],          \
                          (int)ax1[RT][1], 0, 0};                                                                       \
        _Pragma("unroll") for (int c2 = 0; c2 < 2; ++c2)                                                                \
          acc[RT][2 * pr + c2] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, xb[c2], acc[RT][2 * pr + c2],    \
                                                                                  2, 2, RT, sa, 0, sb[c2]);             \
      }
      CF_MX_ROW(0) CF_MX_ROW(1) CF_MX_ROW(2) CF_MX_ROW(3)
#undef CF_MX_ROW
    };
    ldm(am[0], 0, 0);
    ldm(am[1], 0, 1);
    ldx(0);
    // Items per tap: main k-step 0, main k-step 1, cross.  The operands of the item two behind are requested in the MIDDLE
    // of an item (their buffer was freed by the item before), with the only sched_barrier of the item right behind the
    // requests: they cannot sink to their first use, while the LDS reads of the NEXT item's first half may rise above the
    // second half's MFMAs.  (An explicit software pipeline of the B fragments - two buffers, one barrier per 16 MFMAs - was
    // built and measured at the same time, 904 vs 897 us, for 16 more registers: docs/experiments/r5_heads_mx_kernel.md.)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int toff = ((tap / 3) * P_W + tap % 3) * ROWB;
      const int o0 = toff + g * 64, o1 = toff + g * 64 + 16;   // channels 0-31 / 32-63: lane group g holds 8 g .. 8 g + 7 of them
#ifdef CF_MX_ARM_NOA       // dev timing arm (garbage results): no weight stream inside the loop
#define CF_MX_LD(x)
#else
#define CF_MX_LD(x) x
#endif
      main_half(am[0], o0, 0);
      if (tap > 0) CF_MX_LD(ldx(tap));                     // (cross operands of THIS tap: freed by the item before)
      __builtin_amdgcn_sched_barrier(0);
      main_half(am[0], o0, 1);
      main_half(am[1], o1, 0);
      if (tap + 1 < 9) CF_MX_LD(ldm(am[0], tap + 1, 0));
      __builtin_amdgcn_sched_barrier(0);
      main_half(am[1], o1, 1);
      cross_pair(toff, 0);
      cross_pair(toff, 1);
      if (tap + 1 < 9) CF_MX_LD(ldm(am[1], tap + 1, 1));
      __builtin_amdgcn_sched_barrier(0);
      cross_pair(toff, 2);
      cross_pair(toff, 3);
    }
    __builtin_amdgcn_sched_barrier(0);       // (the epilogue's loads stay behind the last item)
    if (PC) {                                // pc_hm on bf16x3: 3 k-steps of 4 taps x 8 channels, weights x 2^(s+4)
      const unsigned char* wp = q.w_first[head] + (size_t)4 * 9 * MX_SLAB + (size_t)wave * (3 * 4 * 2048);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        bf16x8 ph[4], pl[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          ph[rt] = *reinterpret_cast<const bf16x8*>(wp + (i * 4 + rt) * 2048 + (unsigned)lane * 16u);
          pl[rt] = *reinterpret_cast<const bf16x8*>(wp + (i * 4 + rt) * 2048 + 1024 + (unsigned)lane * 16u);
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          bf16x8 xh[4], xl[4];
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            xh[ct] = *reinterpret_cast<const bf16x8*>(xt + rowb[4 * hf + ct] + pc_off[i]);
            xl[ct] = *reinterpret_cast<const bf16x8*>(xt + rowb[4 * hf + ct] + pc_off[i] + 16);
          }
#pragma unroll
          for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
              acc[rt][4 * hf + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl[rt], xh[ct], acc[rt][4 * hf + ct], 0, 0, 0);
              acc[rt][4 * hf + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[rt], xl[ct], acc[rt][4 * hf + ct], 0, 0, 0);
              acc[rt][4 * hf + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[rt], xh[ct], acc[rt][4 * hf + ct], 0, 0, 0);
            }
        }
      }
    }
    return;
  }
  const unsigned char* w1 = q.w_first[head];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][c][r] = 0.0f;

  bf16x8 wh[2][4], wl[2][4];                 // weight fragments one k-step ahead: set ks & 1
  auto load_w = [&](bf16x8 (&dh)[4], bf16x8 (&dl)[4], int ks) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      dh[rt] = *wfrag(w1, wave * 4 + rt, ks, 0, q.n_ks, lane);
      dl[rt] = *wfrag(w1, wave * 4 + rt, ks, 1, q.n_ks, lane);
    }
  };
  load_w(wh[0], wl[0], 0);

#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    if (ks + 1 < NK) load_w(wh[(ks + 1) & 1], wl[(ks + 1) & 1], ks + 1);
    int off, lo;
    if (ks < NKF) {
      const int tap = ks / (NS / 2), half = ks % (NS / 2);
      off = ((tap / 3) * P_W + tap % 3) * ROWB + half * 128 + koff;
      lo = 32;
    } else {
      off = pc_off[ks - NKF < 3 ? ks - NKF : 0];
      lo = 16;
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {         // tile rows 0-3, then 4-7: half of the B fragments live at a time
      bf16x8 xh[4], xl[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        xh[ct] = *reinterpret_cast<const bf16x8*>(xt + rowb[4 * hf + ct] + off);
        xl[ct] = *reinterpret_cast<const bf16x8*>(xt + rowb[4 * hf + ct] + off + lo);
      }
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc[rt][4 * hf + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ks & 1][rt], xh[ct], acc[rt][4 * hf + ct], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc[rt][4 * hf + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks & 1][rt], xl[ct], acc[rt][4 * hf + ct], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc[rt][4 * hf + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks & 1][rt], xh[ct], acc[rt][4 * hf + ct], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  };

  if (HID < 0 ? p.n_hidden > 0 : HID == 1) {   // (the host launches these with hloop == 1)
    const int head = head0;
    f32x4 acc[4][8];
    first_layer(head, acc);
    // hidden layers need all 256 channels of a pixel: the two 64-pixel halves of the tile go through
    // the LDS-resident chain of cf_head_tail one after the other (LDS stays at 66 KiB: 2 workgroups/CU)
    __syncthreads();                         // every wave is done with the patch
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 a2[4][4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) a2[rt][ct] = acc[rt][4 * half + ct];
      store_hidden_tile16<MX>(xt, a2, q.b_first[head], wave, lane, MX ? q.first_scale[head] : 1.0f, c16);
      if constexpr (MX) __builtin_amdgcn_sched_barrier(0);     // (the chain's first weight loads stay behind the tile stores)
      __syncthreads();
      head_tail_from_lds16(p, xt, head, TileMap{b, y0 + (64 >> TSH) * half, x0, q.H, q.W, TSH});
      __syncthreads();
    }
    return;
  }
  if constexpr (HID == 1) return;

  for (int head = head0; head < head1; ++head) {
  f32x4 acc[4][8];
  first_layer(head, acc);
  // ---- output layer from registers: this wave's 64 hidden channels = 2 k-steps of 32
  f32x4 oacc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) oacc[c][r] = 0.0f;
  {
    const float* b1 = q.b_first[head] + wave * 64 + 4 * g;
    const unsigned char* wo = q.w_out_perm[head];
    const float fsc = MX ? q.first_scale[head] : 1.0f;
    (void)fsc;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int ks2 = wave * 2 + s2;
      const bf16x8 ah = *wfrag(wo, 0, ks2, 0, 8, lane), al = *wfrag(wo, 0, ks2, 1, 8, lane);
      const f32x4 ba = *reinterpret_cast<const f32x4*>(b1 + 32 * s2);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(b1 + 32 * s2 + 16);
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) {
        float v[8], hi[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float bj = j < 4 ? ba[j] : bb[j - 4];
          v[j] = fmaxf(MX ? __builtin_fmaf(acc[2 * s2 + (j >> 2)][ct][j & 3], fsc, bj) : acc[2 * s2 + (j >> 2)][ct][j & 3] + bj, 0.0f);
          hi[j] = bf16_rne(v[j]);
        }
        const u32x4 ph = {pack2(hi[0], hi[1]), pack2(hi[2], hi[3]), pack2(hi[4], hi[5]), pack2(hi[6], hi[7])};
        const u32x4 pl = {pack2(v[0] - hi[0], v[1] - hi[1]), pack2(v[2] - hi[2], v[3] - hi[3]),
                          pack2(v[4] - hi[4], v[5] - hi[5]), pack2(v[6] - hi[6], v[7] - hi[7])};
#ifdef CF_MX_ARM_NOCVT   // dev timing arm (garbage results): the accumulator bits as operands, no bias / ReLU / split arithmetic
        const bf16x8 xh = __builtin_bit_cast(bf16x8, u32x4{__builtin_bit_cast(unsigned, acc[2 * s2][ct][0]), __builtin_bit_cast(unsigned, acc[2 * s2][ct][1]),
                                                           __builtin_bit_cast(unsigned, acc[2 * s2][ct][2]), __builtin_bit_cast(unsigned, acc[2 * s2][ct][3])});
        const bf16x8 xl = __builtin_bit_cast(bf16x8, u32x4{__builtin_bit_cast(unsigned, acc[2 * s2 + 1][ct][0]), __builtin_bit_cast(unsigned, acc[2 * s2 + 1][ct][1]),
                                                           __builtin_bit_cast(unsigned, acc[2 * s2 + 1][ct][2]), __builtin_bit_cast(unsigned, acc[2 * s2 + 1][ct][3])});
        (void)ph; (void)pl;
#else
        const bf16x8 xh = __builtin_bit_cast(bf16x8, ph), xl = __builtin_bit_cast(bf16x8, pl);
#endif
        oacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh, oacc[ct], 0, 0, 0);
        oacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl, oacc[ct], 0, 0, 0);
        oacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh, oacc[ct], 0, 0, 0);
      }
    }
  }
  const int n_out = p.n_out[head], act = p.act[head];
#ifdef CF_MX_ARM_NORED     // dev timing arm (garbage results): no partial-sum exchange, no barriers, one store per lane
  {
    float sum = 0.0f;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) sum += (oacc[ct][0] + oacc[ct][1]) + (oacc[ct][2] + oacc[ct][3]);
    if (sum == 12345.0f) p.out[head][tid] = sum;
  }
  continue;
#endif
  float* red = reinterpret_cast<float*>(xt + hp16_patch_bytes(PC));   // [wave][n 16][px 128], BEHIND the patch (which the next head reuses)
#pragma unroll
  for (int ct = 0; ct < 8; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = 4 * g + r;
      if (n < n_out) red[(wave * 16 + n) * HP_PX + ct * 16 + c16] = oacc[ct][r];
    }
  __syncthreads();
  {
    const int px = tid & (HP_PX - 1);
    const int y = y0 + (px >> TSH), x = x0 + (px & TMASK);
    if (y < q.H && x < q.W) {
      const float* bo = p.b_out[head];
      float* out = p.out[head];
      float* out2 = p.out2[head];
      for (int n = tid >> 7; n < n_out; n += 2) {
        const float raw = red[n * HP_PX + px] + red[(16 + n) * HP_PX + px] + red[(32 + n) * HP_PX + px] +
                          red[(48 + n) * HP_PX + px] + bo[n];
        const size_t o = ((size_t)b * n_out + n) * p.HW + (size_t)y * q.W + x;
        float v = raw;
        if (act == CF_ACT_RELU) v = fmaxf(raw, 0.0f);
        else if (act == CF_ACT_SIGMOID_CLAMP) v = fminf(fmaxf(cf_sigmoid(raw), 1e-4f), 1.0f - 1e-4f);
        out[o] = v;
        if (act == CF_ACT_RAW_AND_SIGDEPTH) out2[o] = 1.0f / (cf_sigmoid(raw) + 1e-6f) - 1.0f;
      }
    }
  }
  __syncthreads();                           // the partial sums are consumed: the next head may overwrite them
  }                                          // head loop
}

// ---------------------------------------------------------------------------------------------
// fp32 NHWC feature map -> the 272-byte mx rows head_patch16_kernel<.., MX> stages (layout: there).  One thread per
// (pixel, 32-channel block); the arithmetic is cf_mx.h: mx_pack_block (shared with the DCN epilogue).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_feat_mx_kernel(const float* __restrict__ x, int in_stride,
                                                           unsigned char* __restrict__ rows, long M, float scale) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long m = t >> 1;
  const int blk = (int)(t & 1);
  if (m >= M) return;
  const f32x4* src = reinterpret_cast<const f32x4*>(x + m * in_stride + 32 * blk);
  float v[32];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f32x4 q = src[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * i + e] = q[e];
  }
  mx_pack_block(v, rows + m * 272, blk, scale);
}

}  // namespace

extern "C" int cf_pack_feat_mx_scaled(const float* x, int in_stride, void* rows, long M, float scale, void* stream) {
  CF_REQUIRE(x && rows && M > 0, "cf_pack_feat_mx: null tensor or M=%ld", M);
  const float sc = cf_resolve_in_scale(scale);
  CF_REQUIRE(sc > 0.0f, "cf_pack_feat_mx_scaled: scale must be 0 (= 16) or a power of two");
  CF_REQUIRE(in_stride >= 64 && in_stride % 4 == 0, "cf_pack_feat_mx: in_stride=%d (64 channels, 16-byte aligned rows)", in_stride);
  CF_REQUIRE(M < (1L << 30), "cf_pack_feat_mx: M=%ld too large", M);
  const long threads = 2 * M;
  hipLaunchKernelGGL(pack_feat_mx_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     in_stride, static_cast<unsigned char*>(rows), M, sc);
  return cf_check_launch("cf_pack_feat_mx");
}

extern "C" int cf_pack_feat_mx(const float* x, int in_stride, void* rows, long M, void* stream) {
  return cf_pack_feat_mx_scaled(x, in_stride, rows, M, 16.0f, stream);
}

static int fill_tail(const cf_head_tail_args* a, HeadTailK& k, const char* who, bool need_x) {
  CF_REQUIRE(a != nullptr, "%s: null args", who);
  CF_REQUIRE(!need_x || (a->x && a->x_stride >= 256 && a->x_stride % 8 == 0), "%s: bad input tensor", who);
  CF_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0, "%s: bad geometry", who);
  CF_REQUIRE(a->n_heads >= 1 && a->n_heads <= CF_MAX_HEADS, "%s: n_heads=%d", who, a->n_heads);
  CF_REQUIRE(a->n_hidden >= 0 && a->n_hidden <= 2, "%s: n_hidden=%d", who, a->n_hidden);
  const long M = (long)a->B * a->H * a->W;
  CF_REQUIRE(M < (1L << 31), "%s: tensor too large", who);
  k.x = reinterpret_cast<const unsigned char*>(a->x);
  k.x_stride = a->x_stride;
  k.M = (int)M;
  k.HW = a->H * a->W;
  k.n_heads = a->n_heads;
  k.n_hidden = a->n_hidden;
  for (int i = 0; i < a->n_heads; ++i) {
    for (int l = 0; l < a->n_hidden; ++l) {
      CF_REQUIRE(a->w_hidden[i][l] && a->b_hidden[i][l], "%s: head %d layer %d weights missing", who, i, l);
      k.w_hidden[i][l] = reinterpret_cast<const unsigned char*>(a->w_hidden[i][l]);
      k.b_hidden[i][l] = a->b_hidden[i][l];
    }
    CF_REQUIRE(a->w_out[i] && a->b_out[i] && a->out[i], "%s: head %d output layer missing", who, i);
    CF_REQUIRE(a->n_out[i] >= 1 && a->n_out[i] <= 32, "%s: head %d n_out=%d", who, i, a->n_out[i]);
    CF_REQUIRE(!need_x || (a->c_base[i] >= 0 && a->c_base[i] % 8 == 0 && a->c_base[i] + 256 <= a->x_stride),
               "%s: head %d channel slice out of range", who, i);
    CF_REQUIRE(a->act[i] != CF_ACT_RAW_AND_SIGDEPTH || a->out2[i], "%s: head %d needs out2", who, i);
    k.w_out[i] = reinterpret_cast<const unsigned char*>(a->w_out[i]);
    k.b_out[i] = a->b_out[i];
    k.out[i] = a->out[i];
    k.out2[i] = a->out2[i];
    k.c_base[i] = a->c_base[i];
    k.n_out[i] = a->n_out[i];
    k.act[i] = a->act[i];
  }
  return CF_OK;
}

// Kernels of rounds 1-3 that no default or switchable path of the host dispatches any more - the stand-alone tail
// (cf_head_tail), the slot-table fused head and the 32x32x16 patch kernel (cf_head_fused without mfma16 / layout3x3) - are
// compiled only with -DCF_LEGACY_HEADS (build.py never sets it): the default library answers those calls with CF_EINVAL.
#ifndef CF_LEGACY_HEADS
#define CF_LEGACY_ONLY(what) CF_REQUIRE(false, "%s is a legacy kernel path: rebuild libcfhip with -DCF_LEGACY_HEADS (CF_EXTRA_FLAGS), " \
                                               "or pack the heads for the 16x16x32 patch kernel (mfma16 = 1, layout3x3 = 1, n_out <= 16)", what)
#endif

extern "C" int cf_head_tail(const cf_head_tail_args* a, void* stream) {
  HeadTailK k{};
  const int rc = fill_tail(a, k, "cf_head_tail", true);
  if (rc != CF_OK) return rc;
#ifdef CF_LEGACY_HEADS
  static CfLdsLimit lds_limit;
  lds_limit.ensure(head_tail_kernel, HT_LDS, HT_LDS);
  const int tiles = (k.M + HT_PX - 1) / HT_PX;
  hipLaunchKernelGGL(head_tail_kernel, dim3(tiles * a->n_heads), dim3(256), HT_LDS, (hipStream_t)stream, k);
  return cf_check_launch("cf_head_tail");
#else
  (void)stream;
  CF_LEGACY_ONLY("cf_head_tail");
  return CF_EINVAL;
#endif
}

extern "C" int cf_head_fused(const cf_head_fused_args* a, void* stream) {
  CF_REQUIRE(a != nullptr, "cf_head_fused: null args");
  HeadFusedK k{};
  const int rc = fill_tail(&a->tail, k.t, "cf_head_fused", false);
  if (rc != CF_OK) return rc;
  CF_REQUIRE(a->n_src >= 1 && a->n_src <= 2, "cf_head_fused: n_src=%d", a->n_src);
  for (int i = 0; i < a->n_src; ++i) {
    CF_REQUIRE(a->src[i] && a->src_c[i] > 0 && a->src_c[i] % 8 == 0, "cf_head_fused: source %d invalid", i);
    k.src[i] = reinterpret_cast<const unsigned char*>(a->src[i]);
    k.src_c[i] = a->src_c[i];
  }
  if (a->mx) {                             // the mx operand stream has no slot table: only the 3x3 patch kernel reads it
    CF_REQUIRE(a->layout3x3 && a->mfma16 && a->src_c[0] == 64 && (a->n_src == 1 || a->src_c[1] == 8),
               "cf_head_fused: mx = 1 needs layout3x3 = 1, mfma16 = 1, a 64-channel mx source [and an 8-channel pc_hm source]");
  } else {
    CF_REQUIRE(a->slots && a->K_pad > 0 && a->K_pad % 64 == 0, "cf_head_fused: K_pad=%d must be a multiple of 64", a->K_pad);
    CF_REQUIRE(a->K_pad / 32 <= HF_MAX_CHUNKS, "cf_head_fused: K_pad=%d exceeds %d", a->K_pad, HF_MAX_CHUNKS * 32);
  }
  k.slots = a->slots;
  k.n_chunks = a->K_pad / 32;
  k.H = a->tail.H;
  k.W = a->tail.W;
  for (int i = 0; i < a->tail.n_heads; ++i) {
    CF_REQUIRE(a->w_first[i] && a->b_first[i], "cf_head_fused: head %d first layer missing", i);
    k.w_first[i] = reinterpret_cast<const unsigned char*>(a->w_first[i]);
    k.b_first[i] = a->b_first[i];
  }
  if (a->layout3x3 && a->src_c[0] == 64 && (a->n_src == 1 || a->src_c[1] == 8)) {
    // 2-D patch kernel: K order = 9 taps x 64 feature channels [, then the pc_hm taps pairwise]
    HeadPatchK hp{};
    hp.t = k.t;
    hp.src[0] = k.src[0]; hp.src[1] = k.src[1];
    hp.src_c[0] = k.src_c[0]; hp.src_c[1] = k.src_c[1];
    hp.H = k.H; hp.W = k.W;
    hp.tiles_x = (k.W + HP_TW - 1) / HP_TW;
    hp.tiles_y = (k.H + HP_TH - 1) / HP_TH;
    const bool m16 = a->mfma16 != 0;         // fragments packed for v_mfma_f32_16x16x32_bf16 (k-steps of 32)
    const bool mx = a->mx != 0;              // first layer: fp16 main + FP6 cross terms on the mx feature rows
    CF_REQUIRE(!mx || m16, "cf_head_fused: mx = 1 needs mfma16 = 1 (the tail layers run on 16x16x32 fragments)");
    hp.n_ks = m16 ? a->K_pad / 32 : a->K_pad / 16;
    CF_REQUIRE(mx || a->K_pad / 16 >= (a->n_src == 2 ? 41 : 36), "cf_head_fused: K_pad=%d too small for the 3x3 layout", a->K_pad);
    for (int i = 0; i < a->tail.n_heads; ++i) {
      CF_REQUIRE(!mx || (a->first_scale[i] > 0.0f && a->first_scale[i] < 1e30f), "cf_head_fused: head %d: first_scale missing (mx)", i);
      hp.first_scale[i] = a->first_scale[i];
    }
    if (m16) {
      CF_REQUIRE(mx || (a->K_pad % 32 == 0 && a->K_pad / 32 >= (a->n_src == 2 ? 21 : 18)), "cf_head_fused: K_pad=%d (16x16x32 fragments)", a->K_pad);
      for (int i = 0; i < a->tail.n_heads; ++i)
        CF_REQUIRE(a->tail.n_out[i] <= 16, "cf_head_fused: head %d: n_out=%d > 16 (the 16x16x32 kernels produce ONE 16-row output tile, with or without hidden layers)", i, a->tail.n_out[i]);
    }
    for (int i = 0; i < a->tail.n_heads; ++i) {
      CF_REQUIRE(a->tail.n_hidden > 0 || a->w_out_perm[i], "cf_head_fused: head %d: w_out_perm missing", i);
      hp.w_first[i] = k.w_first[i];
      hp.b_first[i] = k.b_first[i];
      hp.w_out_perm[i] = reinterpret_cast<const unsigned char*>(a->w_out_perm[i]);
    }
    {
      // grid order: head-major (0).  Measured alternatives (CF_HEAD_GROUP = g, dev tools only): tile-major over all heads
      // of the launch takes the same time but 2.2x the fabric fetches - the 7 heads' first-layer weights (4.1 MB) no
      // longer fit an XCD's 4 MB L2 next to the patches; groups of 2 or 4 heads run 3-4 % slower
      const char* e = getenv("CF_HEAD_GROUP");
      hp.group = e ? atoi(e) : 0;
      if (hp.group < 0 || hp.group > a->tail.n_heads) hp.group = 0;
    }
    const int n_groups = hp.group ? (a->tail.n_heads + hp.group - 1) / hp.group : 1;
    const long blocks = (long)hp.tiles_x * hp.tiles_y * a->tail.B * (hp.group ? (long)n_groups * hp.group : a->tail.n_heads);
    CF_REQUIRE(blocks < (1L << 31), "cf_head_fused: grid too large");
#ifdef CF_LEGACY_HEADS
    static CfLdsLimit lim_plain, lim_pc;
    lim_plain.ensure(head_patch_kernel<4, false>, HP_LDS, HP_LDS);
    lim_pc.ensure(head_patch_kernel<4, true>, HP_LDS, HP_LDS);
#endif
    if (m16) {
      // heads without hidden layers: a workgroup walks several heads on one patch (chosen below; CF_HEAD_LOOP overrides
      // for dev tools).  With hidden layers the chain rewrites the patch: one head per workgroup.
      int hloop = 1;
      // tile orientation: 8 x 16 or 16 x 8 pixels, whichever covers the map with fewer tiles (results do not depend on it;
      // CF_HEAD_TILE = 0 / 1 forces one for dev tools)
      const long t_land = (long)((k.W + 15) / 16) * ((k.H + 7) / 8), t_port = (long)((k.W + 7) / 8) * ((k.H + 15) / 16);
      bool portrait = t_port < t_land;
      if (const char* e = getenv("CF_HEAD_TILE")) portrait = atoi(e) != 0;
      if (portrait) {
        hp.tiles_x = (k.W + 7) / 8;
        hp.tiles_y = (k.H + 15) / 16;
      }
      if (a->tail.n_hidden == 0) {
        // heads per workgroup: 1, 2 or half of them, whichever needs the fewest rounds of (2 workgroups per CU) x (heads +
        // a quarter of a head's time for the patch) - small batches want many short workgroups (bs=1: 103 vs 123 us with
        // 1 vs 4 heads), bs=16 the long ones (1307 vs 1331 us).  The results do not depend on it.
        static const int slots = [] {
          int dev = 0, cus = 256;
          if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
          return 2 * (cus > 0 ? cus : 256);
        }();
        const long tiles = (long)hp.tiles_x * hp.tiles_y * a->tail.B;
        const int n = a->tail.n_heads, cand[3] = {1, 2, (n + 1) / 2};
        double best = 0.0;
        for (int i = 0; i < 3; ++i) {
          const int h = cand[i] < 1 ? 1 : (cand[i] > n ? n : cand[i]);
          const double cost = (double)((tiles * ((n + h - 1) / h) + slots - 1) / slots) * (h + 0.25);
          if (i == 0 || cost < best - 1e-9) { best = cost; hloop = h; }
        }
        if (const char* e = getenv("CF_HEAD_LOOP")) hloop = atoi(e);
        if (hloop < 1) hloop = 1;
        if (hloop > n) hloop = n;
      }
      hp.hloop = hloop;
      const long blocks16 = (long)hp.tiles_x * hp.tiles_y * a->tail.B * ((a->tail.n_heads + hloop - 1) / hloop);
      const bool hidden = a->tail.n_hidden > 0;
      const bool pc = a->n_src == 2;
      const int lds = hp16_lds(pc, hidden);
      auto launch = [&](auto kernel, CfLdsLimit& lim) {
        lim.ensure(kernel, lds, hp16_lds(pc, false));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks16), dim3(256), lds, (hipStream_t)stream, hp);
      };
      static CfLdsLimit lim16[4], limx[4];
      if (mx) {
        static CfLdsLimit limxh[2];
        if (hidden) {
          // (hidden layers behind an mx first layer WITHOUT the pc_hm source: that instantiation does not fit 256 registers
          //  without scratch, and no configuration of the reference has such heads - its packing stays bf16x3)
          CF_REQUIRE(pc, "cf_head_fused: mx = 1 with hidden layers needs the pc_hm source (n_src = 2); pack such heads for bf16x3");
          if (portrait) launch(head_patch16_kernel<4, true, true, true, 1>, limxh[1]);
          else launch(head_patch16_kernel<4, true, false, true, 1>, limxh[0]);
        } else {
          if (pc && portrait) launch(head_patch16_kernel<4, true, true, true, 0>, limx[3]);
          else if (pc) launch(head_patch16_kernel<4, true, false, true, 0>, limx[2]);
          else if (portrait) launch(head_patch16_kernel<4, false, true, true, 0>, limx[1]);
          else launch(head_patch16_kernel<4, false, false, true, 0>, limx[0]);
        }
        return cf_check_launch("cf_head_fused");
      }
      if (pc && portrait) launch(head_patch16_kernel<4, true, true>, lim16[3]);
      else if (pc) launch(head_patch16_kernel<4, true, false>, lim16[2]);
      else if (portrait) launch(head_patch16_kernel<4, false, true>, lim16[1]);
      else launch(head_patch16_kernel<4, false, false>, lim16[0]);
      return cf_check_launch("cf_head_fused");
    }
#ifdef CF_LEGACY_HEADS
    if (a->n_src == 2)
      hipLaunchKernelGGL((head_patch_kernel<4, true>), dim3((unsigned)blocks), dim3(256), HP_LDS, (hipStream_t)stream, hp);
    else
      hipLaunchKernelGGL((head_patch_kernel<4, false>), dim3((unsigned)blocks), dim3(256), HP_LDS, (hipStream_t)stream, hp);
    return cf_check_launch("cf_head_fused");
#else
    (void)blocks;
    CF_LEGACY_ONLY("cf_head_fused with 32x32x16 fragments (mfma16 = 0)");
#endif
  }
  // the slot-table kernel reads every weight as 32x32x16 fragments: 16x16x32-packed ones would be misread silently
  CF_REQUIRE(a->mfma16 == 0, "cf_head_fused: mfma16 fragments need the 3x3 patch layout (layout3x3 = 1, 64-channel first "
                             "source [, 8-channel second]); this launch would run on the 32x32x16 slot-table kernel");
#ifdef CF_LEGACY_HEADS
  static CfLdsLimit lds_limit;
  lds_limit.ensure(head_fused_kernel, HF_LDS, HF_LDS);
  const int tiles = (k.t.M + HT_PX - 1) / HT_PX;
  hipLaunchKernelGGL(head_fused_kernel, dim3(tiles * a->tail.n_heads), dim3(256), HF_LDS, (hipStream_t)stream, k);
  return cf_check_launch("cf_head_fused");
#else
  CF_LEGACY_ONLY("cf_head_fused on the slot-table kernel (layout3x3 = 0)");
  return CF_EINVAL;
#endif
}
