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
#include "cf_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

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
  return reinterpret_cast<const bf16x8*>(w + ((((size_t)rt * n_ks + ks) * 2 + plane) * 64 + lane) * 16);
}

// accumulator (lane = pixel ct*32+li, reg r = channel 64w + 32rt + (r&3) + 8(r>>2) + 4h) -> ReLU(acc + b)
// -> split bf16 -> LDS tile [plane][px][528 B]
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

// 4 per-wave partial output tiles [32 n][64 px] -> LDS -> sum + bias + activation -> NCHW
__device__ __forceinline__ void reduce_and_store(const HeadTailK& p, unsigned char* xt, const f32x16 (&oacc)[2],
                                                 int head, int m0) {
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
  const int m = m0 + px;
  if (m < p.M) {
    const int b = m / p.HW, pix = m - b * p.HW;
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
__device__ __forceinline__ void head_tail_from_lds(const HeadTailK& p, unsigned char* xt, int head, int m0) {
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
    bf16x8 wh[2][2], wl[2][2];  // [buffer][rt]
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      wh[0][rt] = *wfrag(w, wave * 2 + rt, 0, 0, 16, lane);
      wl[0][rt] = *wfrag(w, wave * 2 + rt, 0, 1, 16, lane);
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < 16) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          wh[nxt][rt] = *wfrag(w, wave * 2 + rt, ks + 1, 0, 16, lane);
          wl[nxt][rt] = *wfrag(w, wave * 2 + rt, ks + 1, 1, 16, lane);
        }
      }
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
  reduce_and_store(p, xt, oacc, head, m0);
}

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
    head_tail_from_lds(p, xt, head, m0);
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
  reduce_and_store(p, xt, oacc, head, m0);
}

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
  u32x4 sh, sl;
  auto load_b = [&](int c) {
    const cf_slot s = q.slots[c * 4 + su];
    const int src = __builtin_amdgcn_readfirstlane(q.slots[c * 4].src);
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
  // two chunks (four k-steps) per iteration so the four fragment sets are addressed statically
  for (int c = 0; c < q.n_chunks; c += 2) {
    unsigned char* b0 = xt;
    unsigned char* b1 = xt + HF_BUF;
    const int ks = c * 2;
    // chunk c  (buffer 0)
    mma_kstep(b0, 0, wh[0], wl[0]);
    if (ks + 4 < n_ks) load_w(wh[0], wl[0], ks + 4);
    mma_kstep(b0, 1, wh[1], wl[1]);
    if (ks + 5 < n_ks) load_w(wh[1], wl[1], ks + 5);
    store_b(b1);                       // chunk c+1 (requested a chunk ago)
    if (c + 2 < q.n_chunks) load_b(c + 2);
    __syncthreads();
    // chunk c+1 (buffer 1)
    mma_kstep(b1, 0, wh[2], wl[2]);
    if (ks + 6 < n_ks) load_w(wh[2], wl[2], ks + 6);
    mma_kstep(b1, 1, wh[3], wl[3]);
    if (ks + 7 < n_ks) load_w(wh[3], wl[3], ks + 7);
    if (c + 2 < q.n_chunks) {
      store_b(b0);                     // chunk c+2
      if (c + 3 < q.n_chunks) load_b(c + 3);
    }
    __syncthreads();
  }
  // hidden = ReLU(acc + b) -> LDS tile (the staging buffers are dead: last barrier above)
  store_hidden_tile(xt, acc, q.b_first[head], wave, li, h);
  __syncthreads();
  head_tail_from_lds(p, xt, head, m0);
}

}  // namespace

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

extern "C" int cf_head_tail(const cf_head_tail_args* a, void* stream) {
  HeadTailK k{};
  const int rc = fill_tail(a, k, "cf_head_tail", true);
  if (rc != CF_OK) return rc;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(head_tail_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, HT_LDS);
    attr_set = true;
  }
  const int tiles = (k.M + HT_PX - 1) / HT_PX;
  hipLaunchKernelGGL(head_tail_kernel, dim3(tiles * a->n_heads), dim3(256), HT_LDS, (hipStream_t)stream, k);
  return cf_check_launch("cf_head_tail");
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
  CF_REQUIRE(a->slots && a->K_pad > 0 && a->K_pad % 64 == 0, "cf_head_fused: K_pad=%d must be a multiple of 64", a->K_pad);
  k.slots = a->slots;
  k.n_chunks = a->K_pad / 32;
  k.H = a->tail.H;
  k.W = a->tail.W;
  for (int i = 0; i < a->tail.n_heads; ++i) {
    CF_REQUIRE(a->w_first[i] && a->b_first[i], "cf_head_fused: head %d first layer missing", i);
    k.w_first[i] = reinterpret_cast<const unsigned char*>(a->w_first[i]);
    k.b_first[i] = a->b_first[i];
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(head_fused_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, HT_LDS);
    attr_set = true;
  }
  const int tiles = (k.t.M + HT_PX - 1) / HT_PX;
  hipLaunchKernelGGL(head_fused_kernel, dim3(tiles * a->tail.n_heads), dim3(256), HT_LDS, (hipStream_t)stream, k);
  return cf_check_launch("cf_head_fused");
}
