// Host-side weight preparation of the f16x3 operators (SURVEY 8(b) "cf_pack_weights": one-time BN fold + layout), so that a
// host that is not Python can feed cf_conv2d_f16x3 / cf_conv3x3_f16x3 / cf_conv3x3_root_f16x3 / cf_conv3x3_proj_f16x3 and
// cf_dcn_v2_f16x3.  Pure CPU code: host pointers in, host buffers out (the caller copies them to the device once).
// Same arithmetic, operation for operation, as centerfusiondetect3d_amd/packing.py (fold_bn, pack_conv_f16, pack_dcn_f16):
// tests/test_cabi.py compares the bytes.  Replaces what the reference does implicitly by keeping conv + BatchNorm apart
// (model/networks/dla.py:29, 36-39, 151-159) and by torchvision's own weight layout (dla.py:461-470).
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "cf_common.h"

namespace {

// w' = w * g / sqrt(v + eps),  b' = (b - mean) * g / sqrt(v + eps) + beta   (fp32, packing.fold_bn)
void fold(const float* w, const float* b, const cf_pack_bn* bn, int co, long per_out, std::vector<float>& wf, std::vector<float>& bf) {
  wf.assign(w, w + (size_t)co * per_out);
  bf.assign(co, 0.0f);
  if (b) memcpy(bf.data(), b, sizeof(float) * co);
  if (!bn || !bn->gamma) return;
  for (int o = 0; o < co; ++o) {
    const float scale = bn->gamma[o] / sqrtf(bn->var[o] + bn->eps);
    for (long i = 0; i < per_out; ++i) wf[(size_t)o * per_out + i] *= scale;
    bf[o] = (bf[o] - bn->mean[o]) * scale + bn->beta[o];
  }
}

// 2^s with max|w| * 2^s in [2^13, 2^14); the exponent is taken in fp32 as packing.py takes it
int scale_exp(double wmax) {
  if (!(wmax > 0.0)) return 0;
  const float q = (float)(16384.0 / wmax);
  return (int)floorf(log2f(q));
}

// dense (n_pad x k_pad, double) -> fragment order [rt][ks][plane: hi, lo][h][i][8] of fp16
void fragments(const std::vector<double>& w, int n_pad, int k_pad, int s_exp, uint16_t* out) {
  const double sc = ldexp(1.0, s_exp);
  const int n_ks = k_pad / 16;
  for (int n = 0; n < n_pad; ++n)
    for (int k = 0; k < k_pad; ++k) {
      const float ws = (float)(w[(size_t)n * k_pad + k] * sc);
      const _Float16 hi = (_Float16)ws;
      const _Float16 lo = (_Float16)(ws - (float)hi);
      const int rt = n / 32, i = n % 32, ks = k / 16, h = (k % 16) / 8, j = k % 8;
      const size_t base = ((((size_t)rt * n_ks + ks) * 2 + 0) * 64 + (h * 32 + i)) * 8 + j;
      uint16_t bits;
      memcpy(&bits, &hi, 2);
      out[base] = bits;
      memcpy(&bits, &lo, 2);
      out[base + 64 * 8] = bits;
    }
}

struct ConvLayout {
  int n_pad = 0, k_pad = 0, patch = 0;
  std::vector<cf_slot> slots;
  std::vector<int> col;      // per slot: first input channel in the concatenated weight (or -1), with its tap in tap[]
  std::vector<int> tap_r, tap_q;
  std::vector<int> from_proj;
};

int conv_layout(const cf_pack_conv_desc* d, ConvLayout& L) {
  CF_REQUIRE(d && d->weight && d->src && d->n_src >= 1 && d->n_src <= CF_MAX_SRC, "cf_pack_conv_f16x3: bad descriptor");
  CF_REQUIRE(d->cout > 0 && d->kh > 0 && d->kw > 0 && d->stride > 0 && d->dilation > 0, "cf_pack_conv_f16x3: bad geometry");
  const int pad = d->pad < 0 ? (d->kh - 1) / 2 * d->dilation : d->pad;
  L.n_pad = d->cout <= 32 ? 32 : (d->cout + 63) / 64 * 64;
  const cf_pack_src& s0 = d->src[0];
  L.patch = d->kh == 3 && d->kw == 3 && (d->stride == 1 || d->stride == 2) && d->dilation == 1 && pad == 1 && d->n_src == 1 &&
            s0.channels % 16 == 0 && s0.c_base == 0 && s0.stride % 8 == 0;
  auto push = [&](int src, int dy, int dx, int c_off, int col, int r, int q, int proj) {
    L.slots.push_back(cf_slot{src, dy, dx, c_off});
    L.col.push_back(col); L.tap_r.push_back(r); L.tap_q.push_back(q); L.from_proj.push_back(proj);
  };
  if (!L.patch) {
    int c_lo = 0;
    for (int si = 0; si < d->n_src; ++si) {
      const cf_pack_src& s = d->src[si];
      CF_REQUIRE(s.stride % 8 == 0 && s.c_base % 8 == 0 && s.channels % 8 == 0 && s.channels > 0,
                 "cf_pack_conv_f16x3: source %d needs channels, stride and c_base in multiples of 8", si);
      int n_slots = 0;
      for (int r = 0; r < d->kh; ++r)
        for (int q = 0; q < d->kw; ++q)
          for (int g = 0; g < s.channels / 8; ++g, ++n_slots)
            push(si, r * d->dilation - pad, q * d->dilation - pad, s.c_base + 8 * g, c_lo + 8 * g, r, q, 0);
      for (; n_slots % 4; ++n_slots) push(si, 0, 0, -1, -1, 0, 0, 0);
      c_lo += s.channels;
    }
  } else {
    for (int cs = 0; cs < s0.channels / 16; ++cs)
      for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q)
          for (int g = 0; g < 2; ++g) push(0, r - 1, q - 1, 16 * cs + 8 * g, 16 * cs + 8 * g, r, q, 0);
    while (L.slots.size() % 4) push(0, 0, 0, -1, -1, 0, 0, 0);
  }
  if (d->proj_weight) {
    const cf_pack_src& ps = d->proj;
    CF_REQUIRE(L.patch && d->stride == 1 && L.slots.size() % 4 == 0, "cf_pack_conv_f16x3: a projection rides on a slice-major stride-1 3x3 packing");
    CF_REQUIRE(ps.channels > 0 && ps.channels % 32 == 0 && ps.stride % 8 == 0 && ps.c_base % 8 == 0, "cf_pack_conv_f16x3: projected source");
    for (int g = 0; g < ps.channels / 8; ++g) push(1, 0, 0, ps.c_base + 8 * g, 8 * g, 0, 0, 1);
  }
  L.k_pad = (int)L.slots.size() * 8;
  return CF_OK;
}

}  // namespace

extern "C" int cf_pack_conv_f16x3_info(const cf_pack_conv_desc* d, cf_pack_info* info) {
  CF_REQUIRE(info != nullptr, "cf_pack_conv_f16x3_info: null info");
  ConvLayout L;
  const int rc = conv_layout(d, L);
  if (rc != CF_OK) return rc;
  info->n_pad = L.n_pad;
  info->k_pad = L.k_pad;
  info->n_slots = (int32_t)L.slots.size();
  info->patch = L.patch;
  info->out_scale = 0.0f;
  info->weight_bytes = (size_t)L.n_pad * L.k_pad * 2 * sizeof(uint16_t);
  return CF_OK;
}

extern "C" int cf_pack_conv_f16x3(const cf_pack_conv_desc* d, void* weight_out, cf_slot* slots_out, float* bias_out,
                                  cf_pack_info* info) {
  CF_REQUIRE(weight_out && slots_out && bias_out && info, "cf_pack_conv_f16x3: null output");
  ConvLayout L;
  int rc = conv_layout(d, L);
  if (rc != CF_OK) return rc;
  int ci = 0;
  for (int si = 0; si < d->n_src; ++si) ci += d->src[si].channels;
  const long per_out = (long)ci * d->kh * d->kw;
  std::vector<float> wf, bf, pwf, pbf;
  fold(d->weight, d->bias, d->bn.gamma ? &d->bn : nullptr, d->cout, per_out, wf, bf);
  if (d->proj_weight) {
    fold(d->proj_weight, d->proj_bias, d->proj_bn.gamma ? &d->proj_bn : nullptr, d->cout, d->proj.channels, pwf, pbf);
    for (int o = 0; o < d->cout; ++o) bf[o] = bf[o] + pbf[o];
  }
  std::vector<double> w((size_t)L.n_pad * L.k_pad, 0.0);
  double wmax = 0.0;
  for (size_t j = 0; j < L.slots.size(); ++j) {
    if (L.col[j] < 0) continue;
    for (int o = 0; o < d->cout; ++o)
      for (int e = 0; e < 8; ++e) {
        const double v = L.from_proj[j] ? (double)pwf[(size_t)o * d->proj.channels + L.col[j] + e]
                                        : (double)wf[(((size_t)o * ci + L.col[j] + e) * d->kh + L.tap_r[j]) * d->kw + L.tap_q[j]];
        w[(size_t)o * L.k_pad + 8 * j + e] = v;
        const double a = fabs(v);
        if (a > wmax) wmax = a;
      }
  }
  const int s_exp = scale_exp(wmax);
  fragments(w, L.n_pad, L.k_pad, s_exp, static_cast<uint16_t*>(weight_out));
  memcpy(slots_out, L.slots.data(), L.slots.size() * sizeof(cf_slot));
  for (int n = 0; n < L.n_pad; ++n) bias_out[n] = n < d->cout ? bf[n] : 0.0f;
  rc = cf_pack_conv_f16x3_info(d, info);
  info->out_scale = (float)ldexp(1.0, -(s_exp + 4));
  return rc;
}

extern "C" int cf_pack_dcn_f16_info(int cout, int cin, cf_pack_info* info) {
  CF_REQUIRE(info && cout > 0 && cin > 0 && cin % 32 == 0, "cf_pack_dcn_f16: cout=%d cin=%d (cin in multiples of 32)", cout, cin);
  info->n_pad = (cout + 31) / 32 * 32;
  info->k_pad = 9 * cin;
  info->n_slots = 0;
  info->patch = 0;
  info->out_scale = 0.0f;
  info->weight_bytes = (size_t)info->n_pad * info->k_pad * 2 * sizeof(uint16_t);
  return CF_OK;
}

// weight (Cout, Cin, 3, 3) [+ bias, + the BatchNorm behind the DeformConv: dla.py:399-404] -> K order (tap, channel), fragments
extern "C" int cf_pack_dcn_f16(const float* weight, const float* bias, const cf_pack_bn* bn, int cout, int cin,
                               void* weight_out, float* bias_out, cf_pack_info* info) {
  CF_REQUIRE(weight && weight_out && bias_out, "cf_pack_dcn_f16: null buffer");
  int rc = cf_pack_dcn_f16_info(cout, cin, info);
  if (rc != CF_OK) return rc;
  std::vector<float> wf, bf;
  fold(weight, bias, bn && bn->gamma ? bn : nullptr, cout, (long)cin * 9, wf, bf);
  const int n_pad = info->n_pad, k_pad = info->k_pad;
  std::vector<double> w((size_t)n_pad * k_pad, 0.0);
  double wmax = 0.0;
  for (int o = 0; o < cout; ++o)
    for (int c = 0; c < cin; ++c)
      for (int t = 0; t < 9; ++t) {
        const double v = (double)wf[((size_t)o * cin + c) * 9 + t];
        w[(size_t)o * k_pad + (size_t)t * cin + c] = v;
        if (fabs(v) > wmax) wmax = fabs(v);
      }
  const int s_exp = scale_exp(wmax);
  fragments(w, n_pad, k_pad, s_exp, static_cast<uint16_t*>(weight_out));
  for (int n = 0; n < n_pad; ++n) bias_out[n] = n < cout ? bf[n] : 0.0f;
  info->out_scale = (float)ldexp(1.0, -(s_exp + 4));
  return CF_OK;
}
