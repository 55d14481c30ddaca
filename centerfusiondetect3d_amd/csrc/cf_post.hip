// Index-path kernels of the CenterFusion forward: peak top-K (optionally behind the 3x3 equality
// NMS), radar frustum association, detection gather and radar pillar expansion.
//
// These are integer / compare / scatter kernels bounded by HBM and LDS, not by the MFMA pipe; the
// arithmetic that decides integer results (slice bounds, gates, rounding) is written operation for
// operation like the reference's fp32 / fp64 expressions, and this file is compiled with
// -ffp-contract=off so no multiply-add is fused behind our back.
//
// One workgroup per image/frame (top-k: 16 slice workgroups + a merge): the per-image working set (10x112x200 scores = 896 KB, 3x112x200
// radar map = 269 KB) is L2-resident after the first pass, and every ordering decision (top-K
// order, "last painted box wins", "farthest radar point wins") is resolved inside LDS.
#include "cf_common.h"

namespace {

typedef uint32_t u32x4p __attribute__((ext_vector_type(4)));

constexpr int TOPK_THREADS = 256;   // slice kernel
constexpr int TOPK_MERGE_THREADS = 1024;
constexpr int TOPK_CAP = 4096;      // candidate keys kept in LDS (32 KiB)
constexpr int TOPK_SLICES = 16;     // workgroups per image in the slice pass (32: slice pass 43 -> 26 us, but the merge of
                                    // twice as many lists 12 -> 35 us)
constexpr int TOPK_MAXK = 256;      // K <= number of threads of the slice kernel

__device__ __forceinline__ uint32_t f2u(float f) {  // order-preserving float -> uint (-0 and +0 tie, as they compare)
  uint32_t u = __float_as_uint(f);
  u = (u == 0x80000000u) ? 0u : u;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float u2f(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// score of flat element i = (c, y, x) of one image; with NMS: heat * (maxpool3x3(heat) == heat)
template <bool NMS>
__device__ __forceinline__ float peak_value(const float* __restrict__ img, int i, int H, int W) {
  const float v = img[i];
  if (!NMS) return v;
  const int HW = H * W;
  const int c = i / HW, pix = i - c * HW;
  const int y = pix / W, x = pix - y * W;
  const float* pl = img + (size_t)c * HW;
  float m = v;
  const int y0 = y > 0 ? y - 1 : y, y1 = y < H - 1 ? y + 1 : y;
  const int x0 = x > 0 ? x - 1 : x, x1 = x < W - 1 ? x + 1 : x;
  for (int yy = y0; yy <= y1; ++yy)
    for (int xx = x0; xx <= x1; ++xx) m = fmaxf(m, pl[yy * W + xx]);
  return (m == v) ? v : v * 0.0f;
}

// Descending bitonic sort of P (power of two) 64-bit keys in LDS; whole workgroup participates.
__device__ void bitonic_sort_desc(uint64_t* keys, int P) {
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < P; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint64_t a = keys[i], b = keys[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (a < b) : (a > b)) {
            keys[i] = b;
            keys[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
}

// heat * (maxpool3x3(heat) == heat) as its own fully parallel pass (cf_topk_peaks, nms == 2): one
// thread per element, rows of the 3x3 window served by L1/L2.

__global__ __launch_bounds__(256) void nms_kernel(const float* __restrict__ heat, float* __restrict__ out, int H,
                                                  int W, long total) {
  const long N = (long)H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long plane = i / N;
    out[i] = peak_value<true>(heat + plane * N, (int)(i - plane * N), H, W);
  }
}

// (see C below) keys[0..n) unique, n <= blockDim.x * 4: out[r] = the key of rank r (descending) for r < K.
__device__ __forceinline__ void rank_select_desc(const uint64_t* keys, int n, int K, uint64_t* __restrict__ out) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const uint64_t mine = keys[i];
    int rank = 0;
    int j = 0;
    for (; j + 8 <= n; j += 8) {                                    // eight reads in flight (same address across the wave: LDS broadcast)
      uint64_t o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = keys[j + e];
#pragma unroll
      for (int e = 0; e < 8; ++e) rank += o[e] > mine ? 1 : 0;
    }
    for (; j < n; ++j) rank += keys[j] > mine ? 1 : 0;
    if (rank < K) out[rank] = mine;
  }
}

// Pass 1: top-K of one SLICE of one image (TOPK_SLICES workgroups per image), as sorted 64-bit keys
// (order-preserving score bits << 32 | ~flat_index), i.e. ordered by (score desc, flat index asc)
// == (score desc, class asc, pixel asc).  Every element of the image's top-K is in its slice's
// top-K, so pass 2 only has to merge TOPK_SLICES * K keys.
//  A. every thread takes the max of its strided share -> the K-th largest of the 256 local maxima
//     is a lower bound L of the slice's K-th largest element (at least K elements are >= L);
//  B. elements > L are collected into LDS; if fewer than K, the missing ones are the elements == L
//     with the smallest indices, taken by an index-ordered block scan (the common case on real heat
//     maps: the clamp plateau at 1e-4 ties everywhere);
//  C. the candidates are ordered: up to 1024 of them by RANK COUNTING (every key counts the keys above it - the keys
//     are unique - and the ones ranked below K are written straight to their place: one pass over LDS, one barrier,
//     against the ~45 barrier-separated passes of a bitonic sort), more than that (adversarial inputs) by the sort.
// If more than 4096 elements exceed L (adversarial input), the exact K-th value is found by a
// 4 x 8-bit radix select and step B is repeated with it.
// (a device function of a TOPK_THREADS workgroup: slice `slice` of image `img_i` -> out_keys + (img_i * TOPK_SLICES + slice) * K;
//  the kernel below runs one slice per workgroup, topk_fallback_kernel all slices of an image one after the other)
template <bool NMS>
__device__ __forceinline__ void topk_slice_body(const float* __restrict__ heat, int C, int H, int W, int K,
                                                uint64_t* __restrict__ out_keys, int img_i, int slice) {
  __shared__ __attribute__((aligned(16))) uint64_t keys[TOPK_CAP];
  __shared__ uint32_t hist[256];
  __shared__ uint32_t wave_cnt[TOPK_THREADS / 64];
  __shared__ uint32_t s_gt, s_sel[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int HW = H * W, N = C * HW;
  const float* img = heat + (size_t)img_i * N;
  const int len = (N + TOPK_SLICES - 1) / TOPK_SLICES;
  const int lo = min(slice * len, N), hi = min(lo + len, N);
  uint64_t* out = out_keys + ((size_t)img_i * TOPK_SLICES + slice) * K;
  const int n = hi - lo;
  if (n <= K) {  // tiny slice: everything is a candidate
    for (int i = tid; i < TOPK_MAXK; i += TOPK_THREADS)
      keys[i] = i < n ? (((uint64_t)f2u(peak_value<NMS>(img, lo + i, H, W)) << 32) | (uint32_t)(~(uint32_t)(lo + i))) : 0ull;
    __syncthreads();
    bitonic_sort_desc(keys, TOPK_MAXK);
    for (int j = tid; j < K; j += TOPK_THREADS) out[j] = keys[j];
    return;
  }

  // ---- A: lower bound from local maxima
  // (NMS: the 3x3 neighbourhood is only looked at when the raw value could still matter - a
  //  suppressed non-negative element becomes 0 <= raw, so raw <= bound settles it without the 9
  //  loads; negative values, which suppression would RAISE to -0, always take the full path)
  uint32_t lmax = 0;
  {
    int i = lo + tid;
    for (; i + 3 * TOPK_THREADS < hi; i += 4 * TOPK_THREADS) {      // four independent loads in flight
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = img[i + u * TOPK_THREADS];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (!NMS || f2u(v[u]) > lmax || v[u] < 0.0f)
          lmax = max(lmax, NMS ? f2u(peak_value<NMS>(img, i + u * TOPK_THREADS, H, W)) : f2u(v[u]));
    }
    for (; i < hi; i += TOPK_THREADS) {
      const float v = img[i];
      if (!NMS || f2u(v) > lmax || v < 0.0f) lmax = max(lmax, f2u(peak_value<NMS>(img, i, H, W)));
    }
  }
  // K-th largest of the 256 local maxima by rank counting (ties ranked by thread id): one barrier instead of a sort
  uint32_t* lm = reinterpret_cast<uint32_t*>(keys);
  lm[tid] = lmax;
  __syncthreads();
  {
    const u32x4p* l4 = reinterpret_cast<const u32x4p*>(lm);       // (keys[] is 16-byte aligned; eight 16-byte reads in flight)
    int rank = 0;
#pragma unroll 8
    for (int j4 = 0; j4 < TOPK_THREADS / 4; ++j4) {
      const u32x4p o = l4[j4];
#pragma unroll
      for (int e = 0; e < 4; ++e) rank += (o[e] > lmax || (o[e] == lmax && 4 * j4 + e < tid)) ? 1 : 0;
    }
    if (rank == K - 1) s_sel[0] = lmax;
  }
  __syncthreads();
  uint32_t L = s_sel[0];
  __syncthreads();

  // ---- B: collect everything strictly above the bound
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (tid == 0) s_gt = 0;
    __syncthreads();
    {
      auto take = [&](int i, float v) {
        uint32_t u = f2u(v);
        if (NMS && (u > L || v < 0.0f)) u = f2u(peak_value<NMS>(img, i, H, W));
        if (u > L) {
          const uint32_t pos = atomicAdd(&s_gt, 1u);
          if (pos < TOPK_CAP) keys[pos] = ((uint64_t)u << 32) | (uint32_t)(~(uint32_t)i);
        }
      };
      int i = lo + tid;
      for (; i + 3 * TOPK_THREADS < hi; i += 4 * TOPK_THREADS) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = img[i + u * TOPK_THREADS];
#pragma unroll
        for (int u = 0; u < 4; ++u) take(i + u * TOPK_THREADS, v[u]);
      }
      for (; i < hi; i += TOPK_THREADS) take(i, img[i]);
    }
    __syncthreads();
    if (s_gt <= TOPK_CAP) break;
    // exact K-th largest value by radix select (rare path)
    uint32_t prefix = 0, mask = 0, need = K;
    for (int pass = 3; pass >= 0; --pass) {
      const int shift = pass * 8;
      for (int i = tid; i < 256; i += TOPK_THREADS) hist[i] = 0;
      __syncthreads();
      for (int i = lo + tid; i < hi; i += TOPK_THREADS) {
        const uint32_t u = f2u(peak_value<NMS>(img, i, H, W));
        if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        uint32_t acc = 0;
        int bin = 255;
        for (; bin > 0; --bin) {
          if (acc + hist[bin] >= need) break;
          acc += hist[bin];
        }
        s_sel[0] = (uint32_t)bin;
        s_sel[1] = need - acc;
      }
      __syncthreads();
      prefix |= s_sel[0] << shift;
      mask |= 255u << shift;
      need = s_sel[1];
      __syncthreads();
    }
    L = prefix;  // now fewer than K elements are strictly greater
  }
  const int g = (int)s_gt;
  int total = g;
  if (g < K) {
    // index-ordered selection of the (K - g) smallest-index elements equal to L
    const int r = K - g;
    int found = 0;
    for (int base = lo; base < hi && found < r; base += TOPK_THREADS) {
      const int i = base + tid;
      const bool eq = (i < hi) && (!NMS || f2u(img[i]) >= L || img[i] < 0.0f) && (f2u(peak_value<NMS>(img, i, H, W)) == L);
      const unsigned long long bal = __ballot(eq);
      if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(bal);
      __syncthreads();
      int wave_off = 0, all = 0;
      for (int w = 0; w < TOPK_THREADS / 64; ++w) {
        const int cw = (int)wave_cnt[w];
        if (w < wave) wave_off += cw;
        all += cw;
      }
      if (eq) {
        const int rank = found + wave_off + __popcll(bal & ((1ull << lane) - 1ull));
        if (rank < r) keys[g + rank] = ((uint64_t)L << 32) | (uint32_t)(~(uint32_t)i);
      }
      found += all;
      __syncthreads();
    }
    total = K;
  }
  if (total <= 4 * TOPK_THREADS) {
    __syncthreads();
    rank_select_desc(keys, total, K, out);
    return;
  }
  int P = 1;
  while (P < total) P <<= 1;
  for (int i = total + tid; i < P; i += TOPK_THREADS) keys[i] = 0;
  __syncthreads();
  bitonic_sort_desc(keys, P);
  for (int j = tid; j < K; j += TOPK_THREADS) out[j] = keys[j];
}

template <bool NMS>
__global__ __launch_bounds__(TOPK_THREADS) void topk_slice_kernel(const float* __restrict__ heat, int C, int H,
                                                                  int W, int K, uint64_t* __restrict__ out_keys) {
  topk_slice_body<NMS>(heat, C, H, W, K, out_keys, blockIdx.x / TOPK_SLICES, blockIdx.x % TOPK_SLICES);
}

// Pass 1, register-cached form (maps of up to TOPK_RT * TOPK_RE elements per slice - every map of the 448 x 800 configurations):
// 1024 threads per slice, each holds its <= 16 elements in registers, so the map is read ONCE with every load of the slice
// in flight at the same time (the 256-thread kernel above walks the slice twice, four loads per thread at a time: 14 + 14
// dependent rounds of memory latency per launch - 35 us of the chain between the two head launches).  Same algorithm,
// same keys: A. lower bound L = the K-th largest of G GROUP maxima (a group = 1024 / G neighbouring threads: G >= K distinct
// elements, so at least K elements are >= L); B. everything above L into LDS - from the registers; ties / plateaus and the
// adversarial case exactly as above, on the registers; C. rank counting.
constexpr int TOPK_RT = 1024;
constexpr int TOPK_RE = 16;

__global__ __launch_bounds__(TOPK_RT) void topk_slice_reg_kernel(const float* __restrict__ heat, int C, int H, int W, int K,
                                                                 uint64_t* __restrict__ out_keys) {
  __shared__ uint64_t keys[TOPK_CAP];
  __shared__ __attribute__((aligned(16))) uint32_t gmax[256];
  __shared__ uint32_t hist[256];
  __shared__ uint32_t wave_cnt[TOPK_RT / 64];
  __shared__ uint32_t s_gt, s_sel[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = C * H * W;
  const int img_i = blockIdx.x / TOPK_SLICES, slice = blockIdx.x % TOPK_SLICES;
  const float* img = heat + (size_t)img_i * N;
  const int len = (N + TOPK_SLICES - 1) / TOPK_SLICES;
  const int lo = min(slice * len, N), hi = min(lo + len, N);
  uint64_t* out = out_keys + (size_t)blockIdx.x * K;
  const int n = hi - lo;
  if (n <= K) {  // tiny slice: everything is a candidate
    for (int i = tid; i < TOPK_MAXK; i += TOPK_RT)
      keys[i] = i < n ? (((uint64_t)f2u(img[lo + i]) << 32) | (uint32_t)(~(uint32_t)(lo + i))) : 0ull;
    __syncthreads();
    bitonic_sort_desc(keys, TOPK_MAXK);
    for (int j = tid; j < K; j += TOPK_RT) out[j] = keys[j];
    return;
  }
  // element e of this thread = lo + tid + e * TOPK_RT: for a fixed e the threads walk the slice in index order
  uint32_t u[TOPK_RE];
  uint32_t lmax = 0;
#pragma unroll
  for (int e = 0; e < TOPK_RE; ++e) {
    const int i = lo + tid + e * TOPK_RT;
    u[e] = i < hi ? f2u(img[i]) : 0u;          // (0 ranks below every float but one NaN pattern; validity is tested by index)
  }
#pragma unroll
  for (int e = 0; e < TOPK_RE; ++e) lmax = max(lmax, u[e]);
  // ---- A: the K-th largest of the G group maxima (ties ranked by group id).  G = 128 groups of 8 neighbouring threads for
  //      K <= 128, else 256 groups of 4: the rank counting is G^2 compares of VALU work per workgroup - 65,536 of them at
  //      G = 256 were 7 of this kernel's 18 us - and every candidate's count is split over the 1024 / G threads of its group.
  const int P = K <= 128 ? 8 : 4, G = TOPK_RT / P;
  {
    uint32_t m = lmax;
    m = max(m, (uint32_t)__shfl_xor((int)m, 1));
    m = max(m, (uint32_t)__shfl_xor((int)m, 2));
    if (P == 8) m = max(m, (uint32_t)__shfl_xor((int)m, 4));
    if ((tid & (P - 1)) == 0) gmax[tid / P] = m;
  }
  if (tid == 0) s_sel[0] = 0u;                   // (fewer than K non-empty groups - a slice of < P K elements: everything is taken)
  __syncthreads();
  {
    const int c = tid / P, part = tid & (P - 1), per = G / P;      // candidate c against entries [part * per, (part + 1) * per)
    const uint32_t mine = gmax[c];
    const u32x4p* g4 = reinterpret_cast<const u32x4p*>(gmax + part * per);
    int rank = 0;
#pragma unroll 4
    for (int j4 = 0; j4 < per / 4; ++j4) {
      const u32x4p o = g4[j4];
#pragma unroll
      for (int e = 0; e < 4; ++e) rank += (o[e] > mine || (o[e] == mine && part * per + 4 * j4 + e < c)) ? 1 : 0;
    }
    rank += __shfl_xor(rank, 1);
    rank += __shfl_xor(rank, 2);
    if (P == 8) rank += __shfl_xor(rank, 4);
    if (part == 0 && rank == K - 1) s_sel[0] = mine;
  }
  __syncthreads();
  uint32_t L = s_sel[0];
  __syncthreads();

  // ---- B0 (the common case): everything >= the bound - at least K elements - fits the candidate buffer: rank counting over
  //      them orders ties at L by index as well (the key's low word), so no tie pass is needed (it cost 8 barrier-separated
  //      rounds on average: 8 of this kernel's 18 us).  A plateau at L (a clamped map) overflows the buffer and takes B below.
  if (tid == 0) s_gt = 0;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < TOPK_RE; ++e) {
    const int i = lo + tid + e * TOPK_RT;
    if (i < hi && u[e] >= L) {
      const uint32_t pos = atomicAdd(&s_gt, 1u);
      if (pos < TOPK_CAP) keys[pos] = ((uint64_t)u[e] << 32) | (uint32_t)(~(uint32_t)i);
    }
  }
  __syncthreads();
  if (s_gt <= TOPK_CAP) {
    rank_select_desc(keys, (int)s_gt, K, out);
    return;
  }
  __syncthreads();

  // ---- B: collect everything strictly above the bound
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (tid == 0) s_gt = 0;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < TOPK_RE; ++e) {
      const int i = lo + tid + e * TOPK_RT;
      if (i < hi && u[e] > L) {
        const uint32_t pos = atomicAdd(&s_gt, 1u);
        if (pos < TOPK_CAP) keys[pos] = ((uint64_t)u[e] << 32) | (uint32_t)(~(uint32_t)i);
      }
    }
    __syncthreads();
    if (s_gt <= TOPK_CAP) break;
    // exact K-th largest value by radix select (rare path)
    uint32_t prefix = 0, mask = 0, need = K;
    for (int pass = 3; pass >= 0; --pass) {
      const int shift = pass * 8;
      for (int i = tid; i < 256; i += TOPK_RT) hist[i] = 0;
      __syncthreads();
#pragma unroll
      for (int e = 0; e < TOPK_RE; ++e)
        if (lo + tid + e * TOPK_RT < hi && (u[e] & mask) == prefix) atomicAdd(&hist[(u[e] >> shift) & 255u], 1u);
      __syncthreads();
      if (tid == 0) {
        uint32_t acc = 0;
        int bin = 255;
        for (; bin > 0; --bin) {
          if (acc + hist[bin] >= need) break;
          acc += hist[bin];
        }
        s_sel[0] = (uint32_t)bin;
        s_sel[1] = need - acc;
      }
      __syncthreads();
      prefix |= s_sel[0] << shift;
      mask |= 255u << shift;
      need = s_sel[1];
      __syncthreads();
    }
    L = prefix;  // now fewer than K elements are strictly greater
  }
  const int g = (int)s_gt;
  int total = g;
  if (g < K) {
    // index-ordered selection of the (K - g) smallest-index elements equal to L
    const int r = K - g;
    int found = 0;
#pragma unroll 1
    for (int e = 0; e < TOPK_RE && found < r; ++e) {
      const int i = lo + tid + e * TOPK_RT;
      uint32_t ue = 0;
#pragma unroll
      for (int q = 0; q < TOPK_RE; ++q) ue = q == e ? u[q] : ue;       // (register array: a select chain, no scratch)
      const bool eq = i < hi && ue == L;
      const unsigned long long bal = __ballot(eq);
      if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(bal);
      __syncthreads();
      int wave_off = 0, all = 0;
      for (int w = 0; w < TOPK_RT / 64; ++w) {
        const int cw = (int)wave_cnt[w];
        if (w < wave) wave_off += cw;
        all += cw;
      }
      if (eq) {
        const int rank = found + wave_off + __popcll(bal & ((1ull << lane) - 1ull));
        if (rank < r) keys[g + rank] = ((uint64_t)L << 32) | (uint32_t)(~(uint32_t)i);
      }
      found += all;
      __syncthreads();
    }
    total = K;
  }
  __syncthreads();
  rank_select_desc(keys, total, K, out);         // (total <= TOPK_CAP = 4 * TOPK_RT)
}

// Merge of the TOPK_SLICES sorted key lists of one image (keys[] in LDS, list o at keys + o * K; descending, keys unique
// over the image - zero padding of tiny slices excepted, which ties only with itself) as a TREE of pairwise merges that each
// keep the top K: 16 -> 8 -> 4 -> 2 -> 1 lists, ping-pong between keys[0 .. 16 K) and tmp[0 .. 8 K).  An element's rank in
// the merge of its pair = its position + the number of the partner's keys above it (one 7-step binary search): 3,000
// searches over the four levels instead of the 24,000 of ranking every key against all 15 other lists at once (11.5 us as
// its own launch).  -> pointer to the K merged keys (in keys[] or tmp[]).  Whole workgroup; ends with a barrier.
__device__ __forceinline__ const uint64_t* merge_sorted_lists(uint64_t* keys, uint64_t* tmp, int K) {
  uint64_t* src = keys;
  uint64_t* dst = tmp;
  for (int lists = TOPK_SLICES; lists > 1; lists >>= 1) {
    const int total = lists * K;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
      const int sl = i / K, pos = i - sl * K;
      const uint64_t mine = src[i];
      const uint64_t* lst = src + (sl ^ 1) * K;          // the partner list
      int lo = 0, hi = K;                                // first position whose key is NOT above mine
      // (equal keys exist only as zero padding; the odd list of a pair counts them as above, so a tie takes two ranks)
      const bool ge = (sl & 1) != 0;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const uint64_t o = lst[mid];
        if (o > mine || (ge && o == mine)) lo = mid + 1; else hi = mid;
      }
      const int rank = pos + lo;
      if (rank < K) dst[(sl >> 1) * K + rank] = mine;
    }
    __syncthreads();
    uint64_t* t = src; src = dst; dst = t;
  }
  return src;
}

// Pass 2: merge the TOPK_SLICES sorted key lists of one image and emit scores / pixel / class.
__global__ __launch_bounds__(TOPK_MERGE_THREADS) void topk_merge_kernel(const uint64_t* __restrict__ in_keys,
                                                                        int K, int HW, float* __restrict__ scores,
                                                                        int32_t* __restrict__ inds,
                                                                        int32_t* __restrict__ classes) {
  extern __shared__ __attribute__((aligned(16))) uint64_t keys[];      // TOPK_SLICES * K keys + TOPK_SLICES / 2 * K of merge scratch
  const int tid = threadIdx.x, total = TOPK_SLICES * K;
  const uint64_t* src = in_keys + (size_t)blockIdx.x * total;
  for (int i = tid; i < total; i += TOPK_MERGE_THREADS) keys[i] = src[i];
  __syncthreads();
  const uint64_t* best = merge_sorted_lists(keys, keys + total, K);
  for (int j = tid; j < K; j += TOPK_MERGE_THREADS) {
    const uint64_t key = best[j];
    const uint32_t idx = ~(uint32_t)key;
    const int c = (int)(idx / (uint32_t)HW);
    scores[(size_t)blockIdx.x * K + j] = u2f((uint32_t)(key >> 32));
    inds[(size_t)blockIdx.x * K + j] = (int32_t)(idx - (uint32_t)c * HW);
    classes[(size_t)blockIdx.x * K + j] = c;
  }
}

// cf_topk_peaks_if_changed: the NMS'd top-K of image blockIdx.x, start to end in ONE workgroup - but only if the map's checksum
// parts differ from the expected ones (sums[0 .. CK_PARTS) against sums[CK_PARTS .. 2 CK_PARTS)); equal parts: return at once.
// The decoder's guard for peaks that were computed beside the forward (decode.py): the common case costs one near-empty
// launch, the rare one (the caller wrote into the heat map between forward and decode) ~0.3 ms - never a wrong result.
constexpr int CK_PARTS = CF_CHECKSUM_PARTS;

__global__ __launch_bounds__(TOPK_THREADS) void topk_fallback_kernel(const float* __restrict__ heat, int C, int H, int W, int K,
                                                                     uint64_t* __restrict__ keys_ws, float* __restrict__ scores,
                                                                     int32_t* __restrict__ inds, int32_t* __restrict__ classes,
                                                                     const unsigned long long* __restrict__ sums) {
  {
    int diff = 0;
    for (int i = threadIdx.x; i < CK_PARTS; i += TOPK_THREADS) diff |= sums[i] != sums[CK_PARTS + i] ? 1 : 0;
    if (!__syncthreads_or(diff)) return;
  }
  extern __shared__ __attribute__((aligned(16))) uint64_t mkeys[];     // merge: TOPK_SLICES * K keys + TOPK_SLICES / 2 * K of scratch
  const int b = blockIdx.x, tid = threadIdx.x, total = TOPK_SLICES * K;
  for (int sl = 0; sl < TOPK_SLICES; ++sl) {
    topk_slice_body<true>(heat, C, H, W, K, keys_ws, b, sl);
    __syncthreads();                                                   // (the body's LDS is reused by the next slice)
  }
  __threadfence_block();                                               // the lists were written by this workgroup's own threads
  __syncthreads();
  const uint64_t* src = keys_ws + (size_t)b * total;
  for (int i = tid; i < total; i += TOPK_THREADS) mkeys[i] = src[i];
  __syncthreads();
  const uint64_t* best = merge_sorted_lists(mkeys, mkeys + total, K);
  const int HW = H * W;
  for (int j = tid; j < K; j += TOPK_THREADS) {
    const uint64_t key = best[j];
    const uint32_t idx = ~(uint32_t)key;
    const int c = (int)(idx / (uint32_t)HW);
    scores[(size_t)b * K + j] = u2f((uint32_t)(key >> 32));
    inds[(size_t)b * K + j] = (int32_t)(idx - (uint32_t)c * HW);
    classes[(size_t)b * K + j] = c;
  }
}

// ---------------------------------------------------------------------------------------------
// Frustum association
// ---------------------------------------------------------------------------------------------
constexpr int FR_THREADS = 1024;
constexpr int FR_SPLIT = 8;         // workgroups per image (pixel shares of the paint pass)
constexpr int FR_MAXK = 256;
constexpr int FR_NB = 4;          // boxes of a wave whose ROI loads are in flight together (8: slower - 37 vs 29 us on 40 x 30 boxes, 23 vs 21 on 9 x 7)

__device__ __forceinline__ void py_slice(int start, int stop, int n, int& s, int& e) {
  if (start < 0) {
    start += n;
    if (start < 0) start = 0;
  } else if (start > n) {
    start = n;
  }
  if (stop < 0) {
    stop += n;
    if (stop < 0) stop = 0;
  } else if (stop > n) {
    stop = n;
  }
  s = start;
  e = stop;
}

struct FrBox {
  int roi_y0, roi_y1, roi_x0, roi_x1;  // ROI of pc_dep searched for a radar hit
  int p_y0, p_y1, p_x0, p_x1;          // rectangle painted on a hit
  float lo, hi;                        // strict depth gate
  float val[3];                        // painted values (depth/max_dist, vx, vz)
  int found;
};

// slice_keys != nullptr (cf_topk_frustum): the peaks arrive as the TOPK_SLICES sorted key lists of topk_slice_*_kernel and
// are merged HERE, by every workgroup of the image (3 us of redundant LDS work against a launch + 11 us); workgroup 0 of the
// image also writes them out as (scores, inds, classes) when asked.  Otherwise `inds` holds the K peaks (cf_frustum_assoc).
__global__ __launch_bounds__(FR_THREADS) void frustum_kernel(
    const int32_t* __restrict__ inds, int K, const float* __restrict__ depth, const float* __restrict__ wh,
    const float* __restrict__ dim, const float* __restrict__ rot, const float* __restrict__ calib,
    const float* __restrict__ pc_dep, int H, int W, float max_pc_dist, float* __restrict__ pc_hm,
    float* __restrict__ pc_hm_nhwc4, unsigned* __restrict__ pc_hm_split8, const uint64_t* __restrict__ slice_keys,
    float* __restrict__ tk_scores, int32_t* __restrict__ tk_inds, int32_t* __restrict__ tk_classes) {
  __shared__ FrBox box[FR_MAXK];
  __shared__ int s_pix[FR_MAXK];               // pixel of peak i
  __shared__ unsigned long long roi_best[FR_MAXK];   // ROI search: min over the gated returns of (value bits << 32 | position)
  __shared__ short cand[FR_MAXK];              // paint: the boxes that touch this workgroup's rows, last painted first
  __shared__ int s_ncand;
  extern __shared__ __attribute__((aligned(16))) uint64_t fr_keys[];   // slice_keys: TOPK_SLICES * K keys + TOPK_SLICES / 2 * K of merge scratch
  // FR_SPLIT workgroups per image: each rebuilds the (cheap) box table and paints its share of the
  // pixels - the per-pixel "last covering hit" scan is what takes the time
  const int b = blockIdx.x / FR_SPLIT, part = blockIdx.x % FR_SPLIT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int HW = H * W;
  const float* dep_b = depth + (size_t)b * HW;
  const float* wh_b = wh + (size_t)b * 2 * HW;
  const float* dim_b = dim + (size_t)b * 3 * HW;
  const float* rot_b = rot + (size_t)b * 8 * HW;
  const float* cal = calib + (size_t)b * 12;
  const float* pc = pc_dep + (size_t)b * 3 * HW;

  if (slice_keys) {
    const int total = TOPK_SLICES * K;
    const uint64_t* src = slice_keys + (size_t)b * total;
    for (int i = tid; i < total; i += FR_THREADS) fr_keys[i] = src[i];
    __syncthreads();
    const uint64_t* best = merge_sorted_lists(fr_keys, fr_keys + total, K);
    if (tid < K) {
      const uint64_t key = best[tid];
      const uint32_t idx = ~(uint32_t)key;
      const int c = (int)(idx / (uint32_t)HW);
      s_pix[tid] = (int)(idx - (uint32_t)c * HW);
      if (part == 0 && tk_inds) {
        tk_scores[(size_t)b * K + tid] = u2f((uint32_t)(key >> 32));
        tk_inds[(size_t)b * K + tid] = s_pix[tid];
        tk_classes[(size_t)b * K + tid] = c;
      }
    }
  } else if (tid < K) {
    s_pix[tid] = inds[(size_t)b * K + tid];
  }
  // (s_pix[tid] is read by the thread that wrote it)

  // ---- per-box geometry (utils/pointcloud.py:347-381, 397-437, 468-476)
  if (tid < K) {
    const int pix = s_pix[tid];
    const int yi = pix / W, xi = pix - yi * W;
    const float xs = (float)xi + 0.5f, ys = (float)yi + 0.5f;
    float w = wh_b[pix], h = wh_b[HW + pix];
    w = w < 0.0f ? 0.0f : w;
    h = h < 0.0f ? 0.0f : h;
    const float b0 = xs - w / 2.0f, b1 = ys - h / 2.0f, b2 = xs + w / 2.0f, b3 = ys + h / 2.0f;
    const float d = dep_b[pix];
    // get_alpha
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = rot_b[(size_t)i * HW + pix];
    const float idx = r[1] > r[5] ? 1.0f : 0.0f;
    const float a1 = atan2f(r[2], r[3]) + (float)(-0.5 * M_PI);
    const float a2 = atan2f(r[6], r[7]) + (float)(0.5 * M_PI);
    const float alpha = a1 * idx + a2 * (1.0f - idx);
    // getDistanceThresh: yaw, rotated box corners, max(z) - min(z)/2
    const float cx = (b0 + b2) / 2.0f, cy = (b1 + b3) / 2.0f;
    float yaw = alpha + atan2f(cx - cal[2], cal[0]);
    const float PI_F = (float)M_PI, TWO_PI_F = (float)(2.0 * M_PI);
    if (yaw > PI_F) yaw -= TWO_PI_F;
    if (yaw < -PI_F) yaw += TWO_PI_F;
    const float cs = cosf(yaw), sn = sinf(yaw);
    const float dh = dim_b[pix], dw = dim_b[HW + pix], dl = dim_b[2 * HW + pix];
    const float hx = 0.5f * dl, hz = 0.5f * dw;
    const float sx[4] = {hx, hx, -hx, -hx}, sz[4] = {hz, -hz, -hz, hz};
    float zmax = -INFINITY, zmin = INFINITY;
    for (int q = 0; q < 8; ++q) {
      const float yc = q < 4 ? 0.0f : -dh;
      const float z = ((-sn) * sx[q & 3] + 0.0f * yc) + cs * sz[q & 3];
      zmax = fmaxf(zmax, z);
      zmin = fminf(zmin, z);
    }
    const float thr = zmax - zmin / 2.0f;
    FrBox bx;
    py_slice((int)floorf(b1), (int)ceilf(b3) + 1, H, bx.roi_y0, bx.roi_y1);
    py_slice((int)floorf(b0), (int)ceilf(b2) + 1, W, bx.roi_x0, bx.roi_x1);
    bx.hi = d + thr;
    const float t = d - thr;
    bx.lo = t > 0.0f ? t : 0.0f;
    const float wi = 0.3f * (b2 - b0), hi_ = 0.3f * (b3 - b1);
    const int w_min = (int)(cx - wi / 2.0f), w_max = (int)(cx + wi / 2.0f);
    const int h_min = (int)(cy - hi_ / 2.0f), h_max = (int)(cy + hi_ / 2.0f);
    py_slice(h_min, h_max + 1, H, bx.p_y0, bx.p_y1);
    py_slice(w_min, w_max + 2, W, bx.p_x0, bx.p_x1);
    bx.found = 0;
    bx.val[0] = bx.val[1] = bx.val[2] = 0.0f;
    box[tid] = bx;
    roi_best[tid] = ~0ull;
  }
  __syncthreads();

  // ---- nearest gated radar return inside each ROI, row-major first-minimum.  A wave takes FR_NB of its boxes at a time (box
  //      i = g0 + 16 k) and walks their ROIs in rounds of 256 positions each: 4 FR_NB independent loads per lane and round.  The
  //      minimum is taken by ONE LDS atomic per lane and box on a 64-bit key (value bits << 32 | position): gated values are
  //      positive, so their bits order like the values, and among equal values the smallest position - the row-major first
  //      occurrence - wins.  No wave reduction (12 dependent cross-lane steps per box were 10 of this kernel's 32 us).
  constexpr int FR_WAVES = FR_THREADS / 64;
  for (int g0 = wave; g0 < K; g0 += FR_NB * FR_WAVES) {
    int rw[FR_NB], n[FR_NB], y0[FR_NB], x0[FR_NB];
    unsigned magic[FR_NB];                                       // ceil(2^32 / rw): q / rw == umulhi(q, magic) for q < 2^32 / rw (q < 2^16 here)
    float gl[FR_NB], gh[FR_NB];
    unsigned long long best[FR_NB];
    int n_max = 0;
#pragma unroll
    for (int k = 0; k < FR_NB; ++k) {
      const int i = g0 + k * FR_WAVES;
      const FrBox& bx = box[i < K ? i : g0];
      const int w_ = bx.roi_x1 - bx.roi_x0, h_ = bx.roi_y1 - bx.roi_y0;
      rw[k] = w_ > 1 ? w_ : 1;
      magic[k] = rw[k] > 1 ? (unsigned)((0x100000000ull + (unsigned)rw[k] - 1) / (unsigned)rw[k]) : 0u;
      n[k] = (i < K && w_ > 0 && h_ > 0) ? w_ * h_ : 0;
      y0[k] = bx.roi_y0; x0[k] = bx.roi_x0; gl[k] = bx.lo; gh[k] = bx.hi;
      best[k] = ~0ull;
      n_max = max(n_max, n[k]);
    }
    for (int base = 0; base < n_max; base += 256) {
      float v[FR_NB][4];
#pragma unroll
      for (int k = 0; k < FR_NB; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int q = base + lane + 64 * j;
          const int qq = q < n[k] ? q : 0;
          const int qr = rw[k] > 1 ? (int)__umulhi((unsigned)qq, magic[k]) : qq;       // qq / rw
          const float t = pc[(y0[k] + qr) * W + x0[k] + (qq - qr * rw[k])];          // (n[k] == 0: position 0 of a valid ROI origin or of box g0)
          v[k][j] = q < n[k] ? t : 0.0f;
        }
#pragma unroll
      for (int k = 0; k < FR_NB; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (v[k][j] != 0.0f && v[k][j] < gh[k] && v[k][j] > gl[k]) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(v[k][j]) << 32) | (unsigned)(base + lane + 64 * j);
            best[k] = key < best[k] ? key : best[k];
          }
    }
#pragma unroll
    for (int k = 0; k < FR_NB; ++k)
      if (best[k] != ~0ull) atomicMin(&roi_best[g0 + k * FR_WAVES], best[k]);
  }
  __syncthreads();
  if (tid < K) {
    const unsigned long long key = roi_best[tid];
    if (key != ~0ull) {
      const FrBox& bx = box[tid];
      const int rw_ = bx.roi_x1 - bx.roi_x0, pos = (int)(unsigned)key;
      const int yy = bx.roi_y0 + pos / rw_, xx = bx.roi_x0 + pos % rw_;
      box[tid].found = 1;
      box[tid].val[0] = __uint_as_float((unsigned)(key >> 32)) / max_pc_dist;
      box[tid].val[1] = pc[HW + yy * W + xx];
      box[tid].val[2] = pc[2 * HW + yy * W + xx];
    }
  }
  __syncthreads();

  // ---- paint: boxes are drawn in top-k order, so for every pixel the LAST covering hit wins.  This workgroup paints
  //      pixels [p0, p_end): only the hits whose rectangle touches those rows are looked at, last painted first.
  float* hm = pc_hm + (size_t)b * 3 * HW;
  const int p_len = (HW + FR_SPLIT - 1) / FR_SPLIT;
  const int p0 = part * p_len, p_end = min(HW, (part + 1) * p_len);
  if (wave == 0) {
    const int y_lo = p0 / W, y_hi = p_end > p0 ? (p_end - 1) / W : y_lo;
    int n_c = 0;
    for (int base = 0; base < K; base += 64) {
      const int i = K - 1 - (base + lane);
      bool ok = false;
      if (i >= 0) {
        const FrBox& bx = box[i];
        ok = bx.found && bx.p_y0 <= y_hi && bx.p_y1 > y_lo && bx.p_x1 > bx.p_x0 && bx.p_y1 > bx.p_y0;
      }
      const unsigned long long bal = __ballot(ok);
      if (ok) cand[n_c + __popcll(bal & ((1ull << lane) - 1ull))] = (short)i;
      n_c += (int)__popcll(bal);
    }
    if (lane == 0) s_ncand = n_c;
  }
  __syncthreads();
  const int n_cand = s_ncand;
  for (int p = p0 + tid; p < p_end; p += FR_THREADS) {
    const int y = p / W, x = p - y * W;
    float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f;
    for (int ci = 0; ci < n_cand; ++ci) {
      const FrBox& bx = box[cand[ci]];
      if (y >= bx.p_y0 && y < bx.p_y1 && x >= bx.p_x0 && x < bx.p_x1) {
        v0 = bx.val[0];
        v1 = bx.val[1];
        v2 = bx.val[2];
        break;
      }
    }
    hm[p] = v0;
    hm[HW + p] = v1;
    hm[2 * HW + p] = v2;
    if (pc_hm_nhwc4) {
      const f32x4 o = {v0, v1, v2, 0.0f};
      reinterpret_cast<f32x4*>(pc_hm_nhwc4)[(size_t)b * HW + p] = o;
    }
    if (pc_hm_split8) {  // [pixel][hi 8][lo 8] bf16, channels 3..7 zero
      const float h0 = (float)(__bf16)v0, h1 = (float)(__bf16)v1, h2 = (float)(__bf16)v2;
      auto pk = [](float a, float bq) {
        return ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)bq) << 16) |
               __builtin_bit_cast(unsigned short, (__bf16)a);
      };
      unsigned* o = pc_hm_split8 + ((size_t)b * HW + p) * 8;
      o[0] = pk(h0, h1); o[1] = pk(h2, 0.0f); o[2] = 0u; o[3] = 0u;
      o[4] = pk(v0 - h0, v1 - h1); o[5] = pk(v2 - h2, 0.0f); o[6] = 0u; o[7] = 0u;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Detection gather (model/decode.py:40-41, 60-64, 132-172)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void decode_row(const cf_decode_args& a, int t, float* o) {
  const int b = t / a.K;
  const int HW = a.H * a.W;
  const int pix = a.inds[t];
  const int yi = pix / a.W, xi = pix - yi * a.W;
  const float xn = (float)xi / (float)a.W, yn = (float)yi / (float)a.H;
  o[0] = a.scores[t];
  o[1] = (float)a.classes[t];
  o[2] = xn;
  o[3] = yn;
  const float xf = xn * (float)a.out_w, yf = yn * (float)a.out_h;
  float xc, yc;
  if (a.reg) {
    xc = xf + a.reg[((size_t)b * 2 + 0) * HW + pix];
    yc = yf + a.reg[((size_t)b * 2 + 1) * HW + pix];
  } else {
    xc = xf + 0.5f;
    yc = yf + 0.5f;
  }
  const float sx = a.norm2d ? (float)a.out_w : 1.0f, sy = a.norm2d ? (float)a.out_h : 1.0f;
  if (a.wh) {
    float w = a.wh[((size_t)b * 2 + 0) * HW + pix], h = a.wh[((size_t)b * 2 + 1) * HW + pix];
    w = (w < 0.0f ? 0.0f : w) * sx;
    h = (h < 0.0f ? 0.0f : h) * sy;
    o[4] = xc - w / 2.0f;
    o[5] = yc - h / 2.0f;
    o[6] = xc + w / 2.0f;
    o[7] = yc + h / 2.0f;
  } else {
    o[4] = o[5] = o[6] = o[7] = 0.0f;
  }
  auto gather = [&](const float* m, int nc, int off, float s0, float s1) {
    for (int c = 0; c < nc; ++c) {
      float v = m ? m[((size_t)b * nc + c) * HW + pix] : 0.0f;
      if (m) v *= (c == 0 ? s0 : (c == 1 ? s1 : 1.0f));
      o[off + c] = v;
    }
  };
  gather(a.rot, 8, 8, 1.0f, 1.0f);
  gather(a.dim, 3, 16, 1.0f, 1.0f);
  gather(a.amodal, 2, 19, sx, sy);
  gather(a.att, 8, 21, 1.0f, 1.0f);
  gather(a.vel, 3, 29, 1.0f, 1.0f);
  gather(a.depth, 1, 32, 1.0f, 1.0f);
}

__global__ void decode_gather_kernel(cf_decode_args a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.B * a.K) return;
  decode_row(a, t, a.det + (size_t)t * 33);
}

// ---------------------------------------------------------------------------------------------
// 2D -> 3D post-processing of decoded detections (utils/postProcess.py:13-85), one thread per row.
// in : det (B,K,33) [score, cls, cxn, cyn, x1, y1, x2, y2, rot8, dim3, amodal2, att8, vel3, depth]
// out: (B,K,54) [score, cls+1, cx, cy, x1, y1, x2, y2, depth, alpha, dim3, amodal2, att8, vel3,
//                loc3, yaw, box3d 8x3]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void post_row(const float* d, const float* __restrict__ cal,
                                         const float* __restrict__ tinv, float out_w, float out_h, float* o) {
  const float t00 = tinv[0], t01 = tinv[1], t02 = tinv[2], t10 = tinv[3], t11 = tinv[4], t12 = tinv[5];
  auto ax = [&](float x, float y) { return (t00 * x + t01 * y) + t02; };
  auto ay = [&](float x, float y) { return (t10 * x + t11 * y) + t12; };
  o[0] = d[0];
  o[1] = d[1] + 1.0f;
  // 2D boxes back to the source image
  o[4] = ax(d[4], d[5]); o[5] = ay(d[4], d[5]);
  o[6] = ax(d[6], d[7]); o[7] = ay(d[6], d[7]);
  const float depth = d[32];
  o[8] = depth;
  // observation angle (pointcloud.py:207-210)
  const float* r = d + 8;
  const float idx = r[1] > r[5] ? 1.0f : 0.0f;
  const float a1 = atan2f(r[2], r[3]) + (float)(-0.5 * M_PI);
  const float a2 = atan2f(r[6], r[7]) + (float)(0.5 * M_PI);
  const float alpha = a1 * idx + a2 * (1.0f - idx);
  o[9] = alpha;
  const float dh = d[16], dw = d[17], dl = d[18];
  o[10] = dh; o[11] = dw; o[12] = dl;
  o[13] = d[19]; o[14] = d[20];
  for (int i = 0; i < 8; ++i) o[15 + i] = d[21 + i];
  // centre = affine(normalised centre * (W,H) + amodal offset)
  const float px = d[2] * out_w + d[19], py = d[3] * out_h + d[20];
  const float cx = ax(px, py), cy = ay(px, py);
  o[2] = cx; o[3] = cy;
  // unproject (ddd.py:143-166) and yaw (ddd.py:122-140)
  const float z = depth - cal[11];
  const float X = ((cx * depth - cal[3]) - cal[2] * z) / cal[0];
  const float Y = ((cy * depth - cal[7]) - cal[6] * z) / cal[5] + dh / 2.0f;
  float yaw = alpha + atan2f(cx - cal[2], cal[0]);
  const float PI_F = (float)M_PI, TWO_PI_F = (float)(2.0 * M_PI);
  if (yaw > PI_F) yaw -= TWO_PI_F;
  if (yaw < -PI_F) yaw += TWO_PI_F;
  o[26] = X; o[27] = Y; o[28] = z;
  o[29] = yaw;
  // velocity re-projected on the heading
  const float cs = cosf(yaw), sn = sinf(yaw);
  const float V = sqrtf(d[29] * d[29] + d[31] * d[31]);
  o[23] = cs * V; o[24] = d[30]; o[25] = -sn * V;
  // 3D box corners (pointcloud.py:239-296 + ddd.py:8-23); zero if any dimension <= 0
  const bool bad = dh <= 0.0f || dw <= 0.0f || dl <= 0.0f;
  const float hx = 0.5f * dl, hz = 0.5f * dw;
  const float sx[4] = {hx, hx, -hx, -hx}, sz[4] = {hz, -hz, -hz, hz};
  for (int q = 0; q < 8; ++q) {
    const float xc = sx[q & 3], yc = q < 4 ? 0.0f : -dh, zc = sz[q & 3];
    const float bx = ((cs * xc + 0.0f * yc) + sn * zc) + X;
    const float by = ((0.0f * xc + 1.0f * yc) + 0.0f * zc) + Y;
    const float bz = (((-sn) * xc + 0.0f * yc) + cs * zc) + z;
    o[30 + q * 3 + 0] = bad ? 0.0f : bx;
    o[30 + q * 3 + 1] = bad ? 0.0f : by;
    o[30 + q * 3 + 2] = bad ? 0.0f : bz;
  }
}

__global__ void post_process_kernel(const float* __restrict__ det, const float* __restrict__ calib,
                                    const float* __restrict__ tinv, int B, int K, float out_w, float out_h,
                                    float* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * K) return;
  post_row(det + (size_t)t * 33, calib + (size_t)(t / K) * 12, tinv, out_w, out_h, out + (size_t)t * 54);
}

// decode + postProcess in ONE launch: the 33-float row stays in registers between the two steps
// (model/decode.py:10-174 -> utils/postProcess.py:13-85); `a.det` may be NULL when only the final rows
// are wanted.  Same arithmetic as the two kernels above (this file is built without FMA contraction),
// so the fused rows equal cf_decode_gather + cf_post_process bit for bit.
__global__ void decode_post_kernel(cf_decode_args a, const float* __restrict__ calib,
                                   const float* __restrict__ tinv, float* __restrict__ post) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.B * a.K) return;
  float d[33];
  decode_row(a, t, d);
  if (a.det) {
    float* o = a.det + (size_t)t * 33;
#pragma unroll
    for (int i = 0; i < 33; ++i) o[i] = d[i];
  }
  post_row(d, calib + (size_t)(t / a.K) * 12, tinv, (float)a.out_w, (float)a.out_h, post + (size_t)t * 54);
}

// ---------------------------------------------------------------------------------------------
// nuScenes result serialisation (dataset/datasets/nuscenes.py:416-557)
// rows (B*K, 12) f32: [translation xyz (global), size w l h, velocity xy (global), score,
//                      class index 0..9, attribute id 0..8, keep]
// rotation (B*K, 4) f64: pose * cs * R_y(yaw)   (w, x, y, z)
// One thread per detection.  fp32 products of the 4x4 transforms are summed pairwise
// ((p0 + p1) + (p2 + p3)), each operation rounded - the order oracle/serialize_ref.py pins.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot4_pairwise(const float* m, float x, float y, float z, float w) {
  return (m[0] * x + m[1] * y) + (m[2] * z + m[3] * w);
}

__device__ __forceinline__ void quat_mul(const double* a, const double* b, double* o) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

__global__ void serialize_rows_kernel(const float* __restrict__ post, int B, int K,
                                      const float* __restrict__ trans_matrix,
                                      const float* __restrict__ velocity_matrix,
                                      const double* __restrict__ cs_rot, const double* __restrict__ pose_rot,
                                      float* __restrict__ rows, double* __restrict__ rotation) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * K) return;
  const int b = t / K;
  const float* p = post + (size_t)t * 54;
  const float* tm = trans_matrix + (size_t)b * 16;
  const float* vm = velocity_matrix + (size_t)b * 16;
  float* o = rows + (size_t)t * 12;
  const float score = p[0];
  const int cls = (int)p[1] - 1;                          // classIds were shifted to 1..10 by postProcess
  const float dh = p[10], dw = p[11], dl = p[12];
  // size = dimension[[1, 2, 0]] = (w, l, h); location.y -= h in Python floats (fp64), then fp32
  const float lx = p[26], lz = p[28];
  const float ly = (float)((double)p[27] - (double)dh);
  o[0] = dot4_pairwise(tm + 0, lx, ly, lz, 1.0f);
  o[1] = dot4_pairwise(tm + 4, lx, ly, lz, 1.0f);
  o[2] = dot4_pairwise(tm + 8, lx, ly, lz, 1.0f);
  o[3] = dw; o[4] = dl; o[5] = dh;
  o[6] = dot4_pairwise(vm + 0, p[23], p[24], p[25], 0.0f);
  o[7] = dot4_pairwise(vm + 4, p[23], p[24], p[25], 0.0f);
  o[8] = score;
  o[9] = (float)cls;
  // attribute: argmax (first maximum) over the class group's slice of nuscenes_att (nuscenes.py:446-454)
  const float* att = p + 15;
  int attr = 0;
  if (cls == 6 || cls == 7) {                             // motorcycle, bicycle
    attr = (att[1] > att[0] ? 1 : 0) + 1;
  } else if (cls == 5) {                                  // pedestrian
    int m = 2;
    if (att[3] > att[m]) m = 3;
    if (att[4] > att[m]) m = 4;
    attr = (m - 2) + 3;
  } else if (cls >= 0 && cls <= 4) {                      // car, truck, bus, trailer, construction_vehicle
    int m = 5;
    if (att[6] > att[m]) m = 6;
    if (att[7] > att[m]) m = 7;
    attr = (m - 5) + 6;
  }
  o[10] = (float)attr;
  // merge filter of model/progressBar.py:116 / detector.py:437: score > -1 and every dimension > 0
  o[11] = (score > -1.0f && dh > 0.0f && dw > 0.0f && dl > 0.0f) ? 1.0f : 0.0f;
  if (rotation) {
    double* q = rotation + (size_t)t * 4;
    const double half = (double)p[29] / 2.0;
    const double ry[4] = {cos(half), 0.0, sin(half), 0.0};
    double tmp[4];
    if (cs_rot && pose_rot) {
      quat_mul(cs_rot + (size_t)b * 4, ry, tmp);
      quat_mul(pose_rot + (size_t)b * 4, tmp, q);
    } else {
      q[0] = ry[0]; q[1] = ry[1]; q[2] = ry[2]; q[3] = ry[3];
    }
  }
}

// Per-sample merge of the cameras' results + top-N cut (convert_eval_format, nuscenes.py:536-553):
// candidates = kept rows of the sample's frames in (frame order, row order); order = stable sort on
// the score, descending (Python sorts (-score, position)); the best `max_keep` survive.
// One workgroup per sample; rank by counting over the candidate list in LDS.
constexpr int SR_THREADS = 256;
constexpr int SR_MAXC = 6144;   // candidates per sample (6 cameras x K <= 1024)

__global__ __launch_bounds__(SR_THREADS) void serialize_topn_kernel(
    const float* __restrict__ rows, int K, const int32_t* __restrict__ sample_ptr,
    const int32_t* __restrict__ sample_frames, int max_keep, int32_t* __restrict__ order,
    int32_t* __restrict__ counts) {
  __shared__ float s_score[SR_MAXC];
  __shared__ int s_row[SR_MAXC];
  __shared__ int s_wave[SR_THREADS / 64];
  __shared__ int s_base;
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int f0 = sample_ptr[s], f1 = sample_ptr[s + 1];
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int f = f0; f < f1; ++f) {
    const int b = sample_frames[f];
    for (int j0 = 0; j0 < K; j0 += SR_THREADS) {
      const int j = j0 + tid;
      const int row = b * K + j;
      const bool keep = j < K && rows[(size_t)row * 12 + 11] != 0.0f;
      const unsigned long long bal = __ballot(keep);
      if (lane == 0) s_wave[wave] = __popcll(bal);
      __syncthreads();
      int off = s_base;
      for (int w = 0; w < wave; ++w) off += s_wave[w];
      const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
      if (keep && pos < SR_MAXC) {
        s_score[pos] = rows[(size_t)row * 12 + 8];
        s_row[pos] = row;
      }
      __syncthreads();
      if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < SR_THREADS / 64; ++w) tot += s_wave[w];
        s_base += tot;
      }
      __syncthreads();
    }
  }
  const int n = min(s_base, SR_MAXC);
  const int kept = min(n, max_keep);
  if (tid == 0) counts[s] = kept;
  for (int i = tid; i < n; i += SR_THREADS) {
    const float si = s_score[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const float sj = s_score[j];
      rank += (sj > si || (sj == si && j < i)) ? 1 : 0;
    }
    if (rank < max_keep) order[(size_t)s * max_keep + rank] = s_row[i];
  }
  for (int i = kept + tid; i < max_keep; i += SR_THREADS) order[(size_t)s * max_keep + i] = -1;
}

// ---------------------------------------------------------------------------------------------
// Pillar expansion (dataset/generic_dataset.py:738-942, datasets/nuscenes.py:221-263), fp64
// ---------------------------------------------------------------------------------------------
constexpr int PL_THREADS = 256;
constexpr int PL_MAXN = 1024;

struct PlBox {
  int y0, y1, x0, x1;
  float d, vx, vz;
};

// One workgroup per frame.  (1) per-point fp64 geometry -> integer box; (2) LDS rank map: every kept
// point atomicMax-es its depth rank over its rectangle, a wave per point (points are depth-ascending,
// painting order = rank order, so the surviving value of a pixel is that of its highest rank);
// (3) resolve ranks to (depth, vx, vz).  The rank map is processed in row bands that fit LDS.
__global__ __launch_bounds__(PL_THREADS) void pillar_kernel(
    const double* __restrict__ pc_2d, const double* __restrict__ pc_3d, const int32_t* __restrict__ counts,
    int max_n, int n_rows, const double* __restrict__ calib, const double* __restrict__ trans, int H, int W,
    double ph, double pw, double pl, float* __restrict__ pc_dep, uint8_t* __restrict__ keep_mask,
    double* __restrict__ xy_out, int band_rows) {
  extern __shared__ __attribute__((aligned(16))) int rank_map[];  // band_rows * W
  __shared__ PlBox box[PL_MAXN];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = counts[b];
  const double* p2 = pc_2d + (size_t)b * 3 * max_n;
  const double* p3 = pc_3d + (size_t)b * n_rows * max_n;
  const double* cal = calib + (size_t)b * 12;
  const double* m = trans + (size_t)b * 6;

  for (int i = tid; i < max_n; i += PL_THREADS) {
    PlBox bx;
    bx.y0 = bx.y1 = bx.x0 = bx.x1 = 0;
    bx.d = bx.vx = bx.vz = 0.0f;
    bool keep = false;
    double tx = 0.0, ty = 0.0;
    if (i < n) {
      const double u = p2[i], v = p2[max_n + i];
      tx = m[0] * u + m[1] * v + m[2];
      ty = m[3] * u + m[4] * v + m[5];
      keep = (tx < (double)W) && (ty < (double)H) && (0.0 < tx) && (0.0 < ty);
      if (keep) {
        // 8 corners of the (h,w,l) pillar standing on the point, yaw 0; corner offsets are float32
        // in the reference (numpy float32 arrays), the sum with the fp64 location is fp64.
        const double X = p3[i], Y = p3[max_n + i], Z = p3[2 * (size_t)max_n + i];
        const double ox = (double)(float)(0.5 * pl), oz = (double)(float)(0.5 * pw);
        const double oy = (double)(float)(-ph);
        double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
        for (int q = 0; q < 8; ++q) {
          const double cxq = X + ((q & 3) < 2 ? ox : -ox);
          const double cyq = Y + (q < 4 ? 0.0 : oy);
          const double czq = Z + (((q & 3) == 0 || (q & 3) == 3) ? oz : -oz);
          const double pu = ((cal[0] * cxq + cal[1] * cyq) + cal[2] * czq) + cal[3];
          const double pv = ((cal[4] * cxq + cal[5] * cyq) + cal[6] * czq) + cal[7];
          const double pz = ((cal[8] * cxq + cal[9] * cyq) + cal[10] * czq) + cal[11];
          const double uu = pu / pz, vv = pv / pz;
          const double ou = m[0] * uu + m[1] * vv + m[2];
          const double ov = m[3] * uu + m[4] * vv + m[5];
          xmin = fmin(xmin, ou); xmax = fmax(xmax, ou);
          ymin = fmin(ymin, ov); ymax = fmax(ymax, ov);
        }
        const double bw = xmax - xmin, bh = ymax - ymin;
        const double y1 = fmax(ty - bh, 0.0), y2 = ty;
        const double x1 = fmax(tx - bw / 2, 0.0), x2 = fmin(tx + bw / 2, (double)W);
        // np.round == round-half-to-even == rint; then numpy slicing clips to the map
        int iy0 = (int)rint(y1), iy1 = (int)rint(y2), ix0 = (int)rint(x1), ix1 = (int)rint(x2);
        py_slice(iy0, iy1, H, bx.y0, bx.y1);
        py_slice(ix0, ix1, W, bx.x0, bx.x1);
        bx.d = (float)p2[2 * (size_t)max_n + i];
        bx.vx = (float)p3[8 * (size_t)max_n + i];
        bx.vz = (float)p3[9 * (size_t)max_n + i];
      }
    }
    if (!keep) bx.y1 = bx.y0 = 0;
    if (i < PL_MAXN) box[i] = bx;
    if (keep_mask) keep_mask[(size_t)b * max_n + i] = keep ? 1 : 0;
    if (xy_out) {
      xy_out[((size_t)b * 2 + 0) * max_n + i] = tx;
      xy_out[((size_t)b * 2 + 1) * max_n + i] = ty;
    }
  }
  __syncthreads();

  float* out = pc_dep + (size_t)b * 3 * H * W;
  const int HW = H * W;
  for (int r0 = 0; r0 < H; r0 += band_rows) {
    const int r1 = min(r0 + band_rows, H);
    const int cells = (r1 - r0) * W;
    for (int i = tid; i < cells; i += PL_THREADS) rank_map[i] = -1;
    __syncthreads();
    for (int i = wave; i < n; i += PL_THREADS / 64) {
      const PlBox bx = box[i];
      const int ya = max(bx.y0, r0), yb = min(bx.y1, r1);
      const int rw = bx.x1 - bx.x0;
      if (yb <= ya || rw <= 0) continue;
      const int cnt = (yb - ya) * rw;
      for (int q = lane; q < cnt; q += 64) {
        const int yy = ya + q / rw, xx = bx.x0 + q % rw;
        atomicMax(&rank_map[(yy - r0) * W + xx], i);
      }
    }
    __syncthreads();
    for (int i = tid; i < cells; i += PL_THREADS) {
      const int rk = rank_map[i];
      float d = 0.0f, vx = 0.0f, vz = 0.0f;
      if (rk >= 0) {
        d = box[rk].d;
        vx = box[rk].vx;
        vz = box[rk].vz;
      }
      const int p = r0 * W + i;
      out[p] = d;
      out[HW + p] = vx;
      out[2 * HW + p] = vz;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Image side of Detector.pre_process (detector.py:226-234): cv2.warpAffine(INTER_LINEAR, constant
// border 0) of the uint8 camera frame in OpenCV's fixed-point arithmetic (10-bit coordinates, 5-bit
// bilinear fractions, 15-bit weights: oracle/preprocess_ref.py states it), then
// ((v / 255.0 - mean) / std) in float64 -> fp32, HWC -> CHW.  One thread per output pixel.
// (This file is built with -ffp-contract=off: M1*y + M2 must round twice, as it does in OpenCV.)
// ---------------------------------------------------------------------------------------------
struct PreK {
  const uint8_t* src;   // (B, Hs, Ws, 3)
  float* out;           // (B, 3, Hd, Wd)
  double m[6];          // dst -> src map (already inverted)
  double mean[3], stdv[3];
  int B, Hs, Ws, Hd, Wd;
};

__global__ __launch_bounds__(256) void preprocess_kernel(PreK p) {
  const long total = (long)p.B * p.Hd * p.Wd;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % p.Wd);
    const long r = i / p.Wd;
    const int y = (int)(r % p.Hd), b = (int)(r / p.Hd);
    const int X0 = __double2int_rn((p.m[1] * (double)y + p.m[2]) * 1024.0) + 16;
    const int Y0 = __double2int_rn((p.m[4] * (double)y + p.m[5]) * 1024.0) + 16;
    const int X = (X0 + __double2int_rn(p.m[0] * (double)x * 1024.0)) >> 5;
    const int Y = (Y0 + __double2int_rn(p.m[3] * (double)x * 1024.0)) >> 5;
    const int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32;
    const int w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    const uint8_t* img = p.src + (size_t)b * p.Hs * p.Ws * 3;
    const bool y0 = (unsigned)sy < (unsigned)p.Hs, y1 = (unsigned)(sy + 1) < (unsigned)p.Hs;
    const bool x0 = (unsigned)sx < (unsigned)p.Ws, x1 = (unsigned)(sx + 1) < (unsigned)p.Ws;
    const uint8_t* p00 = img + ((size_t)(y0 ? sy : 0) * p.Ws + (x0 ? sx : 0)) * 3;
    const uint8_t* p01 = img + ((size_t)(y0 ? sy : 0) * p.Ws + (x1 ? sx + 1 : 0)) * 3;
    const uint8_t* p10 = img + ((size_t)(y1 ? sy + 1 : 0) * p.Ws + (x0 ? sx : 0)) * 3;
    const uint8_t* p11 = img + ((size_t)(y1 ? sy + 1 : 0) * p.Ws + (x1 ? sx + 1 : 0)) * 3;
    const int m00 = (y0 && x0) ? w00 : 0, m01 = (y0 && x1) ? w01 : 0;
    const int m10 = (y1 && x0) ? w10 : 0, m11 = (y1 && x1) ? w11 : 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int acc = p00[c] * m00 + p01[c] * m01 + p10[c] * m10 + p11[c] * m11;
      const int v = (acc + (1 << 14)) >> 15;
      const double n = ((double)v / 255.0 - p.mean[c]) / p.stdv[c];
      p.out[(((size_t)b * 3 + c) * p.Hd + y) * p.Wd + x] = (float)n;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Radar ingest (detector.py:257-283, nuscenes.py:171-199, pointcloud.py:17-49):
// raw sweep (R rows x N points, float64, camera frame) -> depth gate, y offset, pinhole projection,
// image-border gate, depth order -> the padded pc_2d / pc_3d / counts that cf_pillar_expand consumes.
// One workgroup per frame, one thread per point; the order is a rank by counting over LDS
// (depth, then original index: a stable sort; N <= 1024 so N^2 comparisons are a few microseconds).
// Arithmetic order as oracle/radar_ref.py fixes it (products summed left to right, no FMA).
// ---------------------------------------------------------------------------------------------
constexpr int RI_MAXN = 1024;

__global__ __launch_bounds__(RI_MAXN) void radar_ingest_kernel(
    const double* __restrict__ pc, const int32_t* __restrict__ counts_in, int R, int max_n,
    const double* __restrict__ intr, double width, double height, double max_dist, double z_offset,
    int descending, double* __restrict__ pc_2d, double* __restrict__ pc_3d, int32_t* __restrict__ counts_out) {
  __shared__ double s_depth[RI_MAXN];
  __shared__ int s_keep[RI_MAXN];
  __shared__ int s_total;
  const int b = blockIdx.x, i = threadIdx.x;
  const int n = min(counts_in[b], max_n);
  const double* src = pc + (size_t)b * R * max_n;
  const double* K = intr + (size_t)b * 9;
  if (i == 0) s_total = 0;
  bool keep = false;
  double u = 0.0, v = 0.0, z = 0.0, y = 0.0;
  if (i < n) {
    const double x = src[i];
    y = src[(size_t)max_n + i];
    z = src[(size_t)2 * max_n + i];
    keep = !(max_dist > 0.0) || z <= max_dist;
    if (z_offset != 0.0) y -= z_offset;
    const double px = K[0] * x + K[1] * y + K[2] * z;
    const double py = K[3] * x + K[4] * y + K[5] * z;
    const double pz = K[6] * x + K[7] * y + K[8] * z;
    u = px / pz;
    v = py / pz;
    keep = keep && z > 0.0 && u > 1.0 && u < width - 1.0 && v > 1.0 && v < height - 1.0;
  }
  s_depth[i] = z;
  s_keep[i] = keep ? 1 : 0;
  __syncthreads();
  if (keep) {
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const double dj = s_depth[j];
      rank += (s_keep[j] && (dj < z || (dj == z && j < i))) ? 1 : 0;
    }
    atomicAdd(&s_total, 1);
    s_keep[i] = rank + 1;                  // (every thread has read s_keep[i] only as a flag: still non-zero)
  }
  __syncthreads();
  const int total = s_total;
  if (i == 0) counts_out[b] = total;
  double* o2 = pc_2d + (size_t)b * 3 * max_n;
  double* o3 = pc_3d + (size_t)b * R * max_n;
  if (keep) {
    const int rank = s_keep[i] - 1;
    const int pos = descending ? total - 1 - rank : rank;
    o2[pos] = u;
    o2[(size_t)max_n + pos] = v;
    o2[(size_t)2 * max_n + pos] = z;
    for (int r = 0; r < R; ++r) o3[(size_t)r * max_n + pos] = (r == 1) ? y : src[(size_t)r * max_n + i];
  }
  // zero the padding so the output is fully defined
  for (int j = total + i; j < max_n; j += RI_MAXN) {
    o2[j] = 0.0; o2[(size_t)max_n + j] = 0.0; o2[(size_t)2 * max_n + j] = 0.0;
    for (int r = 0; r < R; ++r) o3[(size_t)r * max_n + j] = 0.0;
  }
}

}  // namespace

extern "C" size_t cf_topk_workspace_bytes(int B, int K) {
  return (size_t)(B > 0 ? B : 0) * TOPK_SLICES * (size_t)(K > 0 ? K : 0) * sizeof(uint64_t);
}

extern "C" size_t cf_topk_workspace_bytes_nms(int B, int C, int H, int W, int K) {
  const size_t map = (size_t)(B > 0 ? B : 0) * (size_t)(C > 0 ? C : 0) * (size_t)(H > 0 ? H : 0) * (size_t)(W > 0 ? W : 0);
  return ((cf_topk_workspace_bytes(B, K) + 255) / 256) * 256 + map * sizeof(float);
}

namespace {

// position-weighted 64-bit checksum of a buffer of 32-bit words, as CK_PARTS partial sums: part[b] = sum over the words block b
// walks (16-byte groups g = b * 1024 + t, + CK_PARTS * 1024, ...) of word[i] * (odd 32-bit weight of i) mod 2^64 - exact integer
// arithmetic, a fixed partition, so two runs over the same bits give the same parts; sensitive to WHERE a value sits (a plain
// sum is blind to swaps).  No atomics, no zeroing: 4,096 same-address atomics took 50 us, the memset was a launch of its own.
__global__ __launch_bounds__(1024) void checksum64_kernel(const uint32_t* __restrict__ x, long n, unsigned long long* __restrict__ parts) {
  unsigned long long acc = 0ull;
  const long step = (long)CK_PARTS * 1024;
  const long n4 = (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? n / 4 : 0;           // 16-byte loads over the aligned body
  const u32x4p* x4 = reinterpret_cast<const u32x4p*>(x);
  for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < n4; i += step) {
    const u32x4p v = x4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      acc += (unsigned long long)v[e] * (unsigned long long)(((uint32_t)(4 * i + e) * 2654435761u) | 1u);
  }
  for (long i = 4 * n4 + (long)blockIdx.x * 1024 + threadIdx.x; i < n; i += step)
    acc += (unsigned long long)x[i] * (unsigned long long)(((uint32_t)i * 2654435761u) | 1u);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += (unsigned long long)__shfl_xor((long long)acc, o, 64);
  __shared__ unsigned long long part[16];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0ull;
    for (int w = 0; w < 16; ++w) t += part[w];
    parts[blockIdx.x] = t;
  }
}

}  // namespace

extern "C" int cf_checksum64(const void* x, long n_words, unsigned long long* parts, void* stream) {
  CF_REQUIRE(x && parts && n_words >= 0, "cf_checksum64: null buffer or n_words=%ld", n_words);
  hipLaunchKernelGGL(checksum64_kernel, dim3(CK_PARTS), dim3(1024), 0, (hipStream_t)stream, static_cast<const uint32_t*>(x), n_words, parts);
  return cf_check_launch("cf_checksum64");
}

namespace {

int topk_peaks_impl(const float* heat, int B, int C, int H, int W, int K, int nms, float* scores,
                    int32_t* inds, int32_t* classes, void* workspace, void* stream) {
  CF_REQUIRE(heat && scores && inds && classes && workspace, "cf_topk_peaks: null buffer");
  CF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "cf_topk_peaks: bad geometry");
  CF_REQUIRE(K >= 1 && K <= TOPK_MAXK, "cf_topk_peaks: K=%d outside [1,%d]", K, TOPK_MAXK);
  CF_REQUIRE((long)C * H * W >= K, "cf_topk_peaks: fewer than K elements per image");
  CF_REQUIRE((long)C * H * W < (1L << 31), "cf_topk_peaks: image too large");
  hipStream_t st = (hipStream_t)stream;
  uint64_t* keys = static_cast<uint64_t*>(workspace);
  CF_REQUIRE(nms >= 0 && nms <= 2, "cf_topk_peaks: nms=%d", nms);
  const bool two_pass = nms == 2;
  if (nms == 2) {   // suppressed map first (scratch behind the keys), then the plain top-K over it
    float* sup = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + ((cf_topk_workspace_bytes(B, K) + 255) / 256) * 256);
    const long total = (long)B * C * H * W;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(nms_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st, heat, sup, H, W, total);
    heat = sup;
    nms = 0;
  }
  static const int reg_off = [] { const char* e = getenv("CF_TOPK_REG"); return e ? atoi(e) == 0 : 0; }();   // (dev A/B: CF_TOPK_REG=0 = the 256-thread kernel)
  const long slice_len = ((long)C * H * W + TOPK_SLICES - 1) / TOPK_SLICES;
  if (nms)
    hipLaunchKernelGGL(topk_slice_kernel<true>, dim3(B * TOPK_SLICES), dim3(TOPK_THREADS), 0, st, heat, C, H, W, K, keys);
  // (the register-cached kernel - 1024 threads and 35 KB of LDS per workgroup - for the plain top-K, which sits alone on the chip
  //  between the two head launches; behind the NMS pass - the decoder's top-K, issued BESIDE the secondary head launch - the
  //  256-thread kernel, whose workgroups find room next to the head workgroups: the big ones waited for the whole launch)
  else if (slice_len <= (long)TOPK_RT * TOPK_RE && !reg_off && !two_pass)
    hipLaunchKernelGGL(topk_slice_reg_kernel, dim3(B * TOPK_SLICES), dim3(TOPK_RT), 0, st, heat, C, H, W, K, keys);
  else
    hipLaunchKernelGGL(topk_slice_kernel<false>, dim3(B * TOPK_SLICES), dim3(TOPK_THREADS), 0, st, heat, C, H, W, K, keys);
  const size_t merge_lds = (size_t)(TOPK_SLICES + TOPK_SLICES / 2) * K * sizeof(uint64_t);
  static CfLdsLimit merge_limit;
  merge_limit.ensure(topk_merge_kernel, merge_lds, 65536);
  hipLaunchKernelGGL(topk_merge_kernel, dim3(B), dim3(TOPK_MERGE_THREADS), merge_lds, st, keys, K, H * W, scores, inds, classes);
  return cf_check_launch("cf_topk_peaks");
}

}  // namespace

extern "C" int cf_topk_peaks(const float* heat, int B, int C, int H, int W, int K, int nms, float* scores,
                             int32_t* inds, int32_t* classes, void* workspace, void* stream) {
  return topk_peaks_impl(heat, B, C, H, W, K, nms, scores, inds, classes, workspace, stream);
}

extern "C" int cf_topk_peaks_if_changed(const float* heat, int B, int C, int H, int W, int K, int nms, float* scores,
                                        int32_t* inds, int32_t* classes, void* workspace,
                                        const unsigned long long* sums, void* stream) {
  CF_REQUIRE(heat && scores && inds && classes && workspace && sums, "cf_topk_peaks_if_changed: null buffer");
  CF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "cf_topk_peaks_if_changed: bad geometry");
  CF_REQUIRE(K >= 1 && K <= TOPK_MAXK, "cf_topk_peaks_if_changed: K=%d outside [1,%d]", K, TOPK_MAXK);
  CF_REQUIRE((long)C * H * W >= K && (long)C * H * W < (1L << 31), "cf_topk_peaks_if_changed: image of %ld elements", (long)C * H * W);
  CF_REQUIRE(nms == 1 || nms == 2, "cf_topk_peaks_if_changed: nms=%d (the guard of the decoder's NMS'd peaks: 1 or 2, same result)", nms);
  const size_t lds = (size_t)(TOPK_SLICES + TOPK_SLICES / 2) * K * sizeof(uint64_t);
  static CfLdsLimit limit;
  limit.ensure(topk_fallback_kernel, lds, 65536);
  hipLaunchKernelGGL(topk_fallback_kernel, dim3(B), dim3(TOPK_THREADS), lds, (hipStream_t)stream, heat, C, H, W, K,
                     static_cast<uint64_t*>(workspace), scores, inds, classes, sums);
  return cf_check_launch("cf_topk_peaks_if_changed");
}

extern "C" int cf_frustum_assoc(const int32_t* inds, int K, const float* depth, const float* wh,
                                const float* dim, const float* rot, const float* calib, const float* pc_dep,
                                int B, int H, int W, float max_pc_dist, float* pc_hm, float* pc_hm_nhwc4,
                                void* pc_hm_split8, void* stream) {
  CF_REQUIRE(inds && depth && wh && dim && rot && calib && pc_dep && pc_hm, "cf_frustum_assoc: null buffer");
  CF_REQUIRE(K >= 1 && K <= FR_MAXK, "cf_frustum_assoc: K=%d outside [1,%d]", K, FR_MAXK);
  CF_REQUIRE(B > 0 && H > 0 && W > 0, "cf_frustum_assoc: bad geometry");
  hipLaunchKernelGGL(frustum_kernel, dim3(B * FR_SPLIT), dim3(FR_THREADS), 0, (hipStream_t)stream, inds, K, depth, wh, dim,
                     rot, calib, pc_dep, H, W, max_pc_dist, pc_hm, pc_hm_nhwc4, static_cast<unsigned*>(pc_hm_split8),
                     (const uint64_t*)nullptr, (float*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
  return cf_check_launch("cf_frustum_assoc");
}

// The chain between the two head launches in TWO launches instead of three: slice top-K of the raw heat map, then the
// frustum kernel, which merges the slices' lists in its prologue (the merge launch and its 11 us are gone).
extern "C" int cf_topk_frustum(const float* heat, int C, int K, const float* depth, const float* wh, const float* dim,
                               const float* rot, const float* calib, const float* pc_dep, int B, int H, int W,
                               float max_pc_dist, float* pc_hm, float* pc_hm_nhwc4, void* pc_hm_split8, float* scores,
                               int32_t* inds, int32_t* classes, void* workspace, void* stream) {
  CF_REQUIRE(heat && depth && wh && dim && rot && calib && pc_dep && pc_hm && workspace, "cf_topk_frustum: null buffer");
  CF_REQUIRE((scores && inds && classes) || (!scores && !inds && !classes), "cf_topk_frustum: scores / inds / classes go together");
  CF_REQUIRE(K >= 1 && K <= FR_MAXK && K <= TOPK_MAXK, "cf_topk_frustum: K=%d outside [1,%d]", K, FR_MAXK);
  CF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "cf_topk_frustum: bad geometry");
  CF_REQUIRE((long)C * H * W >= K && (long)C * H * W < (1L << 31), "cf_topk_frustum: image of %ld elements", (long)C * H * W);
  hipStream_t st = (hipStream_t)stream;
  uint64_t* keys = static_cast<uint64_t*>(workspace);
  static const int reg_off = [] { const char* e = getenv("CF_TOPK_REG"); return e ? atoi(e) == 0 : 0; }();
  const long slice_len = ((long)C * H * W + TOPK_SLICES - 1) / TOPK_SLICES;
  if (slice_len <= (long)TOPK_RT * TOPK_RE && !reg_off)
    hipLaunchKernelGGL(topk_slice_reg_kernel, dim3(B * TOPK_SLICES), dim3(TOPK_RT), 0, st, heat, C, H, W, K, keys);
  else
    hipLaunchKernelGGL(topk_slice_kernel<false>, dim3(B * TOPK_SLICES), dim3(TOPK_THREADS), 0, st, heat, C, H, W, K, keys);
  const size_t lds = (size_t)(TOPK_SLICES + TOPK_SLICES / 2) * K * sizeof(uint64_t);     // <= 48 KB at K = 256, + 17 KB static
  static CfLdsLimit fr_limit;
  fr_limit.ensure(frustum_kernel, lds, 65536);
  hipLaunchKernelGGL(frustum_kernel, dim3(B * FR_SPLIT), dim3(FR_THREADS), lds, st, (const int32_t*)nullptr, K, depth, wh, dim,
                     rot, calib, pc_dep, H, W, max_pc_dist, pc_hm, pc_hm_nhwc4, static_cast<unsigned*>(pc_hm_split8),
                     (const uint64_t*)keys, scores, inds, classes);
  return cf_check_launch("cf_topk_frustum");
}

extern "C" int cf_decode_gather(const cf_decode_args* a, void* stream) {
  CF_REQUIRE(a && a->scores && a->inds && a->classes && a->det, "cf_decode_gather: null buffer");
  CF_REQUIRE(a->B > 0 && a->K > 0 && a->H > 0 && a->W > 0, "cf_decode_gather: bad geometry");
  const int n = a->B * a->K;
  hipLaunchKernelGGL(decode_gather_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, *a);
  return cf_check_launch("cf_decode_gather");
}

extern "C" int cf_post_process(const float* det, const float* calib, const float* trans_inv, int B, int K,
                               int out_h, int out_w, float* out, void* stream) {
  CF_REQUIRE(det && calib && trans_inv && out, "cf_post_process: null buffer");
  CF_REQUIRE(B > 0 && K > 0 && out_h > 0 && out_w > 0, "cf_post_process: bad geometry");
  const int n = B * K;
  hipLaunchKernelGGL(post_process_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, det, calib,
                     trans_inv, B, K, (float)out_w, (float)out_h, out);
  return cf_check_launch("cf_post_process");
}

extern "C" int cf_decode_post(const cf_decode_args* a, const float* calib, const float* trans_inv, float* post,
                             void* stream) {
  CF_REQUIRE(a && a->scores && a->inds && a->classes && calib && trans_inv && post, "cf_decode_post: null buffer");
  CF_REQUIRE(a->B > 0 && a->K > 0 && a->H > 0 && a->W > 0 && a->out_h > 0 && a->out_w > 0,
             "cf_decode_post: bad geometry");
  const int n = a->B * a->K;
  hipLaunchKernelGGL(decode_post_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, *a, calib,
                     trans_inv, post);
  return cf_check_launch("cf_decode_post");
}

extern "C" int cf_serialize_nuscenes(const cf_serialize_args* a, void* stream) {
  CF_REQUIRE(a && a->post && a->trans_matrix && a->velocity_matrix && a->rows, "cf_serialize_nuscenes: null buffer");
  CF_REQUIRE(a->B > 0 && a->K > 0 && a->K <= 1024, "cf_serialize_nuscenes: bad geometry (K <= 1024)");
  CF_REQUIRE((a->cs_rot == nullptr) == (a->pose_rot == nullptr), "cf_serialize_nuscenes: cs_rot and pose_rot go together");
  const int n = a->B * a->K;
  hipLaunchKernelGGL(serialize_rows_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, a->post, a->B,
                     a->K, a->trans_matrix, a->velocity_matrix, a->cs_rot, a->pose_rot, a->rows, a->rotation);
  if (a->n_samples > 0) {
    CF_REQUIRE(a->sample_ptr && a->sample_frames && a->order && a->counts, "cf_serialize_nuscenes: null sample tables");
    CF_REQUIRE(a->max_per_sample >= 1, "cf_serialize_nuscenes: max_per_sample=%d", a->max_per_sample);
    hipLaunchKernelGGL(serialize_topn_kernel, dim3(a->n_samples), dim3(SR_THREADS), 0, (hipStream_t)stream, a->rows,
                       a->K, a->sample_ptr, a->sample_frames, a->max_per_sample, a->order, a->counts);
  }
  return cf_check_launch("cf_serialize_nuscenes");
}

extern "C" int cf_serialize_max_candidates(void) { return SR_MAXC; }

extern "C" int cf_pillar_expand(const double* pc_2d, const double* pc_3d, const int32_t* counts, int B,
                                int max_n, int n_rows, const double* calib, const double* trans, int H, int W,
                                double pillar_h, double pillar_w, double pillar_l, float* pc_dep,
                                uint8_t* keep_mask, double* xy_out, void* stream) {
  CF_REQUIRE(pc_2d && pc_3d && counts && calib && trans && pc_dep, "cf_pillar_expand: null buffer");
  CF_REQUIRE(B > 0 && H > 0 && W > 0, "cf_pillar_expand: bad geometry");
  CF_REQUIRE(max_n >= 1 && max_n <= PL_MAXN, "cf_pillar_expand: max_n=%d outside [1,%d]", max_n, PL_MAXN);
  CF_REQUIRE(n_rows >= 10, "cf_pillar_expand: pc_3d needs >= 10 rows (8 = vx, 9 = vz)");
  int band_rows = H;
  const int max_cells = (96 * 1024) / 4;  // 96 KiB of rank map per workgroup
  if ((long)band_rows * W > max_cells) band_rows = max_cells / W;
  CF_REQUIRE(band_rows >= 1, "cf_pillar_expand: W=%d too wide", W);
  const size_t lds = (size_t)band_rows * W * sizeof(int);
  static CfLdsLimit lds_limit;
  lds_limit.ensure(pillar_kernel, lds, 65536);
  hipLaunchKernelGGL(pillar_kernel, dim3(B), dim3(PL_THREADS), lds, (hipStream_t)stream, pc_2d, pc_3d, counts,
                     max_n, n_rows, calib, trans, H, W, pillar_h, pillar_w, pillar_l, pc_dep, keep_mask, xy_out,
                     band_rows);
  return cf_check_launch("cf_pillar_expand");
}

extern "C" int cf_preprocess_images(const uint8_t* src, int B, int Hs, int Ws, const double* map_dst_to_src,
                                    const float* mean, const float* stdv, int Hd, int Wd, float* out, void* stream) {
  CF_REQUIRE(src && map_dst_to_src && mean && stdv && out, "cf_preprocess_images: null buffer");
  CF_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "cf_preprocess_images: bad geometry");
  CF_REQUIRE(Hs < 32768 && Ws < 32768 && Hd < 32768 && Wd < 32768, "cf_preprocess_images: image too large for the fixed-point map");
  PreK k{};
  k.src = src; k.out = out;
  for (int i = 0; i < 6; ++i) k.m[i] = map_dst_to_src[i];        // host memory: six doubles
  for (int c = 0; c < 3; ++c) {
    CF_REQUIRE(stdv[c] != 0.0f, "cf_preprocess_images: std[%d] is zero", c);
    k.mean[c] = (double)mean[c];
    k.stdv[c] = (double)stdv[c];
  }
  k.B = B; k.Hs = Hs; k.Ws = Ws; k.Hd = Hd; k.Wd = Wd;
  const long total = (long)B * Hd * Wd;
  const long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0,
                     (hipStream_t)stream, k);
  return cf_check_launch("cf_preprocess_images");
}

extern "C" int cf_radar_ingest(const double* pc, const int32_t* counts_in, int B, int n_rows, int max_n,
                               const double* intrinsics, int img_w, int img_h, double max_dist, double z_offset,
                               int descending, double* pc_2d, double* pc_3d, int32_t* counts_out, void* stream) {
  CF_REQUIRE(pc && counts_in && intrinsics && pc_2d && pc_3d && counts_out, "cf_radar_ingest: null buffer");
  CF_REQUIRE(B > 0 && n_rows >= 3 && img_w > 2 && img_h > 2, "cf_radar_ingest: bad geometry");
  CF_REQUIRE(max_n >= 1 && max_n <= RI_MAXN, "cf_radar_ingest: max_n=%d outside [1,%d]", max_n, RI_MAXN);
  hipLaunchKernelGGL(radar_ingest_kernel, dim3(B), dim3(RI_MAXN), 0, (hipStream_t)stream, pc, counts_in, n_rows,
                     max_n, intrinsics, (double)img_w, (double)img_h, max_dist, z_offset, descending, pc_2d, pc_3d,
                     counts_out);
  return cf_check_launch("cf_radar_ingest");
}
