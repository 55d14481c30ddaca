"""Oracle-side restatement of the heads' first-layer arithmetic "f16 main term + block-scaled FP6 cross terms".

TEST INFRASTRUCTURE - see oracle/__init__.py.  Restates in float64 / numpy what `head_patch16_kernel<.., MX>` and
`pack_feat_mx_kernel` (centerfusiondetect3d_amd/csrc/cf_heads.hip) compute for the 3x3 layer of
/root/reference/src/lib/model/networks/detectHeads.py:59-79 (and :165-191 for the radar heads):

    W * 2^s = Wh + Wl,   x * 16 = xh + xl        (Wh, xh: fp16 RNE; Wl, xl: the exact fp32 remainders)
    y ~ ( Wh . xh                                  v_mfma_f32_16x16x32_f16: exact products, fp32 accumulation
        + q6(Wh) . q6(xl) + q6(Wl) . q6(xh) ) * 2^-(s+4) + b      one v_mfma_scale_f32_16x16x128_f8f6f4 per tap

q6 = OCP MX FP6 e2m3 (sign, 2 exponent bits of bias 1, 3 mantissa bits: 0, 0.125 .. 7.5) with one E8M0 scale per block
of 32 channels of one tap; the block exponent is the smallest e with max|v| <= 7.5 * 2^e.  The three pc_hm channels of
the radar heads stay on bf16x3 (hi.hi + hi.lo + lo.hi).  Nothing here imports the product package: the quantiser, the
bit packing and the row layout are written out independently (numpy) so that tests/test_gpu_ops.py can compare the
kernels' bytes with them.
"""
import numpy as np
import torch
import torch.nn.functional as F

ASCALE = 16.0
ROW = 272        # bytes per pixel of the mx feature map: 4 segments of [8 fp16 | 8 fp16 | 24 B of FP6 fields + 8 pad], 4 scale bytes, pad (feat_rows_ref)

_E2M3 = np.array([(m / 8.0 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 1)) for e in range(4) for m in range(8)])


def block_exponent(amax):
    """numpy float32 array of block maxima -> int32 exponents: smallest e with amax <= 7.5 * 2^e, never below -127; 0 -> -127."""
    a = np.ascontiguousarray(amax, dtype=np.float32)
    bits = a.view(np.int32)
    e = ((bits >> 23) & 0xFF) - 127 - 2 + ((bits & 0x7FFFFF) > 0x700000)
    return np.maximum(np.where(a == 0, -127, e), -127).astype(np.int32)   # (maxima below 2^-125: a zero block, code 0)


def e2m3_codes(t):
    """float64 array (value / block scale) -> uint8 codes, nearest with ties to the even code, saturating; sign bit from
    signbit(t)."""
    a = np.minimum(np.abs(t), 7.5)
    d = np.abs(a[..., None] - _E2M3)                           # (.., 32)
    best = d.min(-1, keepdims=True)
    cand = d == best                                           # one or two adjacent codes
    first = cand.argmax(-1)
    two = cand.sum(-1) > 1
    code = np.where(two & (first % 2 == 1), first + 1, first)  # tie: the even one of (first, first + 1)
    return (code | (np.signbit(t).astype(np.int64) << 5)).astype(np.uint8)


def e2m3_values(codes):
    c = codes.astype(np.int64)
    return np.where(c & 32, -1.0, 1.0) * _E2M3[c & 31]


def quant_blocks(v):
    """v (..., 32 n) float32/64 -> (codes (..., n, 32) uint8, exponents (..., n) int32, dequantised (..., 32 n) float64)."""
    b = np.asarray(v, dtype=np.float64).reshape(*v.shape[:-1], v.shape[-1] // 32, 32)
    e = block_exponent(np.abs(b).max(-1).astype(np.float32))
    s = np.ldexp(1.0, e)[..., None]
    # a block whose exponent sits at the E8M0 floor (max below 1.875 * 2^-125, fp32 denormals included) is a ZERO block: the
    # pack kernel converts it with scale 1, every field comes out 0 (cf_mx.h)
    codes = np.where((e <= -127)[..., None], np.uint8(0), e2m3_codes(b / s))
    return codes, e, (e2m3_values(codes) * s).reshape(v.shape)


def pack_fields(codes):
    """(..., 32) codes -> (..., 24) bytes, element j at bits [6j, 6j + 6) little endian."""
    out = np.zeros(codes.shape[:-1] + (24,), np.uint8)
    for j in range(32):
        bit = 6 * j
        v = codes[..., j].astype(np.uint32) << (bit % 8)
        out[..., bit // 8] |= (v & 0xFF).astype(np.uint8)
        if bit % 8 > 2:
            out[..., bit // 8 + 1] |= (v >> 8).astype(np.uint8)
    return out


def split_f16(v):
    """float32 numpy v -> (hi, lo): hi = fp16 RNE of clamp(v), lo = v - hi exactly."""
    v = np.clip(np.asarray(v, np.float32), -65504.0, 65504.0)
    hi = v.astype(np.float16).astype(np.float32)
    return hi, v - hi


def feat_rows_ref(feat_nhwc, scale=ASCALE):
    """(M, 64) float32 -> (M, ROW) uint8: what cf_pack_feat_mx (scale 16) / cf_pack_feat_mx_scaled write."""
    x = np.ascontiguousarray(feat_nhwc, dtype=np.float32) * np.float32(scale)
    hi, lo = split_f16(x)
    M = x.shape[0]
    rows = np.zeros((M, ROW), np.uint8)
    h16 = np.ascontiguousarray(np.clip(x, -65504.0, 65504.0).astype(np.float16)).view(np.uint8).reshape(M, 8, 16)   # 8-channel chunks
    for g in range(4):                                         # segment g: chunk g (channels 8g..), chunk 4 + g (channels 32 + 8g..)
        rows[:, 64 * g:64 * g + 16] = h16[:, g]
        rows[:, 64 * g + 16:64 * g + 32] = h16[:, 4 + g]
    hc, he, _ = quant_blocks(hi)
    lc, le, _ = quant_blocks(lo)
    lf, hf = pack_fields(lc), pack_fields(hc)                  # (M, 2, 24)
    for blk in range(2):
        rows[:, 64 * blk + 32:64 * blk + 56] = lf[:, blk]
        rows[:, 64 * (2 + blk) + 32:64 * (2 + blk) + 56] = hf[:, blk]
        rows[:, 256 + blk] = (le[:, blk] + 127).astype(np.uint8)
        rows[:, 258 + blk] = (he[:, blk] + 127).astype(np.uint8)
    return rows


def rows_unpack(rows):
    """(M, ROW) uint8 -> (hi fp16 values (M, 64), q6(lo) (M, 64), q6(hi) (M, 64)) as float64: the row format read back"""
    M = rows.shape[0]
    hi = np.zeros((M, 64))
    for g in range(4):
        hi[:, 8 * g:8 * g + 8] = rows[:, 64 * g:64 * g + 16].copy().view(np.float16)
        hi[:, 32 + 8 * g:40 + 8 * g] = rows[:, 64 * g + 16:64 * g + 32].copy().view(np.float16)
    out = []
    for first, sc in ((0, 256), (2, 258)):
        q = np.zeros((M, 64))
        for blk in range(2):
            seg = rows[:, 64 * (first + blk) + 32:64 * (first + blk) + 56]
            bits = [int.from_bytes(bytes(r), "little") for r in seg]
            codes = np.array([[(b >> (6 * j)) & 63 for j in range(32)] for b in bits], np.uint8)
            q[:, 32 * blk:32 * blk + 32] = e2m3_values(codes) * np.ldexp(1.0, rows[:, sc + blk].astype(np.int32) - 127)[:, None]
        out.append(q)
    return hi, out[0], out[1]


def weight_scale_exp(w):
    wmax = float(w.abs().max())
    return int(torch.floor(torch.log2(torch.tensor(16384.0 / wmax)))) if wmax > 0 else 0


def _bf16_split(v):
    hi = v.to(torch.bfloat16).double()
    lo = (v.double() - hi).float().to(torch.bfloat16).double()
    return hi, lo


def first_layer_mx(feat, pc_hm, weight, bias, scale=ASCALE):
    """feat (B,64,H,W) f32, pc_hm (B,3,H,W) f32 or None, weight (Co, 64 [+3], 3, 3), bias (Co) -> (B,Co,H,W) float64:
    ReLU is NOT applied.  Exact products, float64 accumulation (the MFMA's fp32 accumulation is not modelled).
    scale: the feature map's power-of-two pre-scale (16 unless the model was calibrated for a larger range)."""
    B, C, H, W = feat.shape
    co = weight.shape[0]
    s = weight_scale_exp(weight)
    wf = (weight[:, :64].double() * 2.0 ** s).float().permute(0, 2, 3, 1).reshape(co, 9 * 64).numpy()     # k = (tap, channel)
    wh, wl = split_f16(wf)
    _, _, wh6 = quant_blocks(wh)
    _, _, wl6 = quant_blocks(wl)
    cols = F.unfold(feat.float() * float(scale), 3, padding=1).view(B, C, 9, H * W).permute(0, 2, 1, 3).reshape(B, 9 * 64, H * W)
    out = np.zeros((B, co, H * W))
    for b in range(B):
        x = cols[b].numpy().T                                  # (P, 576), blocks of 32 along k never straddle taps
        xh, xl = split_f16(x)
        _, _, xh6 = quant_blocks(xh)
        _, _, xl6 = quant_blocks(xl)
        acc = wh.astype(np.float64) @ xh.astype(np.float64).T + wh6 @ xl6.T + wl6 @ xh6.T
        out[b] = acc
    y = torch.from_numpy(out).view(B, co, H, W)
    if pc_hm is not None:
        wp = (weight[:, 64:].double() * 2.0 ** s * float(scale)).float()
        ph, pl = _bf16_split(wp)
        xh, xl = _bf16_split(pc_hm.float())
        conv = lambda a, w_: F.conv2d(a, w_, None, 1, 1)
        y = y + conv(xh, ph) + conv(xl, ph) + conv(xh, pl)
    return y * (2.0 ** -s / float(scale)) + bias.double().view(1, -1, 1, 1)
