"""One-time weight preparation for the HIP kernels (host side, runs at model load).

  * eval-mode BatchNorm is folded into the producing conv:  w' = w * g/sqrt(v+eps),
    b' = (b - mean) * g/sqrt(v+eps) + beta   (reference: dla.py:29,36-39,151-159 run BN as a
    separate op; SURVEY.md Appendix B.12 for the DeformConv bias case);
  * weights are re-laid-out as the K-contiguous [N_pad][K_pad] matrix the implicit GEMM reads, in
    "slot" order: one slot = 4 consecutive input channels of one source tensor at one filter tap
    (include/cf_hip.h: cf_slot).  A Root's channel concat (dla.py:35) or the feat||pc_hm concat of
    the secondary heads (fusionModules.py:33) becomes several sources - nothing is concatenated
    in memory.
"""
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

BN_EPS = 1e-5
BK = 32


def fold_bn(weight, bias, bn):
    """bn = (gamma, beta, mean, var) or None -> (weight', bias') float32 on CPU."""
    w = weight.detach().float().cpu()
    co = w.shape[0]
    b = torch.zeros(co) if bias is None else bias.detach().float().cpu()
    if bn is None:
        return w, b
    g, beta, mean, var = (t.detach().float().cpu() for t in bn)
    # (the square root through float64: correctly rounded in fp32, which torch's vectorised fp32 sqrt is not for every
    #  length - the C packer, cf_pack_conv_f16x3 / cf_pack_dcn_f16, uses sqrtf and has to produce the same bytes)
    scale = g / torch.sqrt((var + BN_EPS).double()).float()
    return w * scale.view(-1, *([1] * (w.dim() - 1))), (b - mean) * scale + beta


@dataclass
class Source:
    channels: int      # real channels this source contributes to the conv's Cin
    stride: int        # floats per pixel of the NHWC tensor holding it
    c_base: int = 0    # first channel inside that tensor


@dataclass
class PackedConv:
    weight: torch.Tensor          # [N_pad, K_pad] f32
    bias: torch.Tensor            # [N_pad] f32
    slots: torch.Tensor           # [K_pad/4, 4] int32 (src, dy, dx, c_off)
    n: int
    n_pad: int
    k_pad: int
    kh: int
    stride: int
    pad: int
    real_cin: tuple = ()      # real input channels per source (for FLOP accounting)
    out_scale: float = 0.0    # cf_conv2d_f16x3: 2^-(s+4)
    patch: bool = False       # slice-major 3x3 packing: cf_conv3x3_f16x3 may run it
    proj_k: int = 0           # channels of the 1x1 projection packed behind a slice-major 3x3 (cf_conv3x3_proj_f16x3)

    def to(self, device):
        self.weight = self.weight.to(device).contiguous()
        self.bias = self.bias.to(device).contiguous()
        self.slots = self.slots.to(device).contiguous()
        return self


def pack_conv(weight, bias, sources: Sequence[Source], stride=1, pad=None, dilation=1,
              n_pad: Optional[int] = None) -> PackedConv:
    """weight (Cout, sum(src.channels), kh, kw), BN already folded."""
    co, ci, kh, kw = weight.shape
    assert ci == sum(s.channels for s in sources), (ci, [s.channels for s in sources])
    pad = (kh - 1) // 2 * dilation if pad is None else pad
    n_pad = n_pad or ((co + 31) // 32) * 32
    slots: List[List[int]] = []
    cols = []      # per slot: (cin index of its first channel or -1, number of real channels)
    c_lo = 0
    for si, s in enumerate(sources):
        per_tap = (s.channels + 3) // 4
        n_slots = 0
        for r in range(kh):
            for q in range(kw):
                for g in range(per_tap):
                    real = min(4, s.channels - 4 * g)
                    slots.append([si, r * dilation - pad, q * dilation - pad, s.c_base + 4 * g])
                    cols.append((c_lo + 4 * g, real, r, q))
                    n_slots += 1
        while n_slots % (BK // 4):       # keep every 32-wide chunk inside one source
            slots.append([si, 0, 0, -1])
            cols.append((-1, 0, 0, 0))
            n_slots += 1
        c_lo += s.channels
    k_pad = len(slots) * 4
    w = torch.zeros(n_pad, k_pad)
    wf = weight.float()
    for j, (c0, real, r, q) in enumerate(cols):
        if real:
            w[:co, 4 * j:4 * j + real] = wf[:, c0:c0 + real, r, q]
    b = torch.zeros(n_pad)
    b[:co] = bias
    return PackedConv(w, b, torch.tensor(slots, dtype=torch.int32), co, n_pad, k_pad, kh, stride, pad,
                      tuple(s.channels for s in sources))


def pack_conv_bf16(weight, bias, sources: Sequence[Source], stride=1, pad=None, dilation=1,
                   fragments=False) -> PackedConv:
    """Packing for cf_conv2d_bf16x3: slots of 8 channels, weights [N_pad][2][K_pad] bf16 with
    w = hi + lo (hi = rne(w), lo = rne(w - hi)).  Source.stride = channels per plane."""
    co, ci, kh, kw = weight.shape
    assert ci == sum(s.channels for s in sources), (ci, [s.channels for s in sources])
    pad = (kh - 1) // 2 * dilation if pad is None else pad
    n_pad = ((co + 31) // 32) * 32
    slots, cols, c_lo = [], [], 0
    for si, s in enumerate(sources):
        assert s.stride % 8 == 0 and s.c_base % 8 == 0
        per_tap = (s.channels + 7) // 8
        n_slots = 0
        for r in range(kh):
            for q in range(kw):
                for g in range(per_tap):
                    real = min(8, s.channels - 8 * g)
                    slots.append([si, r * dilation - pad, q * dilation - pad, s.c_base + 8 * g])
                    cols.append((c_lo + 8 * g, real, r, q))
                    n_slots += 1
        while n_slots % 4:                 # a 32-wide chunk = 4 slots, one source per chunk
            slots.append([si, 0, 0, -1])
            cols.append((-1, 0, 0, 0))
            n_slots += 1
        c_lo += s.channels
    while len(slots) % 8:                  # whole 64-deep steps for the BK = 64 main loop
        slots.append([len(sources) - 1, 0, 0, -1])
        cols.append((-1, 0, 0, 0))
    k_pad = len(slots) * 8
    w = torch.zeros(n_pad, k_pad)
    wf = weight.float()
    for j, (c0, real, r, q) in enumerate(cols):
        if real:
            w[:co, 8 * j:8 * j + real] = wf[:, c0:c0 + real, r, q]
    b = torch.zeros(n_pad)
    b[:co] = bias
    if fragments == 16:    # cf_head_fused with mfma16: 16x16x32 fragments
        wt = pack_fragments16(w)
    elif fragments:        # cf_head_fused: A-operand fragment order instead of [N][2][K]
        wt = pack_fragments(w)
    else:
        hi = w.to(torch.bfloat16)
        lo = (w - hi.float()).to(torch.bfloat16)
        wt = torch.stack([hi, lo], dim=1).contiguous()
    return PackedConv(wt, b, torch.tensor(slots, dtype=torch.int32), co, n_pad, k_pad, kh, stride, pad,
                      tuple(s.channels for s in sources))


def pack_conv_f16(weight, bias, sources: Sequence[Source], stride=1, pad=None, dilation=1, proj=None) -> PackedConv:
    """Packing for cf_conv2d_f16x3: fp32 NHWC sources, 8-channel slots (4 per 32-deep chunk), weights
    scaled by 2^s (max|w| -> [2^13, 2^14)), split into fp16 hi/lo and laid out in MFMA A-operand
    fragment order.  PackedConv.out_scale = 2^-(s+4) undoes the weight and activation scales.

    A 3x3 / pad 1 convolution (stride 1 or 2) of ONE source with C % 16 == 0 is packed SLICE-MAJOR
    (k = (16-channel slice, tap, channel)): that is the order cf_conv3x3_f16x3 (LDS patch reuse)
    consumes, and since the slot table spells the same order out the generic kernel runs the very same
    weights (PackedConv.patch marks them).

    proj = (weight (Cout, Cp, 1, 1), bias (Cout,), Source): a 1x1 convolution of a SECOND tensor at the output
    resolution whose products are summed into the same accumulators - a BasicBlock's conv2 together with the Tree's
    `project` of the pooled input, which the reference adds as conv2's residual (dla.py:96-107, 56-62).  Its k-steps
    follow the 3x3 part (slots: source 1, tap (0, 0)), the biases add, one 2^s covers both weight sets;
    PackedConv.proj_k = Cp.  cf_conv3x3_proj_f16x3 runs it, the generic slot kernel runs the same table."""
    co, ci, kh, kw = weight.shape
    assert ci == sum(s.channels for s in sources), (ci, [s.channels for s in sources])
    pad = (kh - 1) // 2 * dilation if pad is None else pad
    n_pad = 32 if co <= 32 else ((co + 63) // 64) * 64
    slots, cols, c_lo = [], [], 0
    patch = (kh == 3 and kw == 3 and stride in (1, 2) and dilation == 1 and pad == 1 and len(sources) == 1
             and sources[0].channels % 16 == 0 and sources[0].c_base == 0 and sources[0].stride % 8 == 0)
    for si, s in enumerate(sources if not patch else ()):
        assert s.stride % 8 == 0 and s.c_base % 8 == 0 and s.channels % 8 == 0, "f16x3 sources need C % 8 == 0"
        n_slots = 0
        for r in range(kh):
            for q in range(kw):
                for g in range(s.channels // 8):
                    slots.append([si, r * dilation - pad, q * dilation - pad, s.c_base + 8 * g])
                    cols.append((c_lo + 8 * g, r, q))
                    n_slots += 1
        while n_slots % 4:
            slots.append([si, 0, 0, -1])
            cols.append((-1, 0, 0))
            n_slots += 1
        c_lo += s.channels
    if patch:
        for cs in range(sources[0].channels // 16):
            for r in range(3):
                for q in range(3):
                    for g in range(2):
                        slots.append([0, r - 1, q - 1, 16 * cs + 8 * g])
                        cols.append((16 * cs + 8 * g, r, q))
        while len(slots) % 4:
            slots.append([0, 0, 0, -1])
            cols.append((-1, 0, 0))
    if proj is not None:
        pw, pb, ps = proj
        assert patch and stride == 1 and len(slots) % 4 == 0, "a projection rides on a slice-major stride-1 3x3 packing"
        assert tuple(pw.shape) == (co, ps.channels, 1, 1) and ps.channels % 32 == 0 and ps.stride % 8 == 0 and ps.c_base % 8 == 0
        slots = slots + [[1, 0, 0, ps.c_base + 8 * g] for g in range(ps.channels // 8)]
        cols = cols + [(8 * g, -1, -1) for g in range(ps.channels // 8)]
        bias = bias + pb
    k_pad = len(slots) * 8
    w = torch.zeros(n_pad, k_pad, dtype=torch.float64)
    wf = weight.double()
    for j, (c0, r, q) in enumerate(cols):
        if c0 >= 0 and r < 0:
            w[:co, 8 * j:8 * j + 8] = proj[0].double()[:, c0:c0 + 8, 0, 0]
        elif c0 >= 0:
            w[:co, 8 * j:8 * j + 8] = wf[:, c0:c0 + 8, r, q]
    wmax = float(w.abs().max())
    s_exp = int(torch.floor(torch.log2(torch.tensor(16384.0 / wmax)))) if wmax > 0 else 0
    ws = (w * 2.0 ** s_exp).float()
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    planes = torch.stack([hi, lo], 0)
    f = planes.view(2, n_pad // 32, 32, k_pad // 16, 2, 8).permute(1, 3, 0, 4, 2, 5).contiguous()
    b = torch.zeros(n_pad)
    b[:co] = bias
    pc = PackedConv(f.view(n_pad // 32, k_pad // 16, 2, 64, 8), b, torch.tensor(slots, dtype=torch.int32),
                    co, n_pad, k_pad, kh, stride, pad, tuple(s.channels for s in sources))
    pc.out_scale = 2.0 ** -(s_exp + 4)
    pc.patch = patch
    if proj is not None:
        pc.real_cin = (sources[0].channels, proj[2].channels)
        pc.proj_k = proj[2].channels
    return pc


def pack_fragments(weight2d, n_pad=None, acc_order=False):
    """(N, K) fp32 -> MFMA A-operand fragment order for cf_head_tail:
    uint8 view of [N_pad/32][K/16][2 (hi, lo)][64 lanes][8 bf16]; lane (i = l & 31, h = l >> 5)
    holds W[32 rt + i][16 ks + 8 h + j], j = 0..7.
    acc_order: position 8h + j of every 16-group holds channel 4h + (j & 3) + 8 (j >> 2) instead - the
    order in which a 32x32 accumulator's register group presents its rows (cf_head_fused w_out_perm)."""
    n, k = weight2d.shape
    assert k % 16 == 0
    n_pad = n_pad or ((n + 31) // 32) * 32
    w = torch.zeros(n_pad, k)
    w[:n] = weight2d.float()
    if acc_order:
        perm = torch.tensor([16 * g + 4 * hh + (j & 3) + 8 * (j >> 2)
                             for g in range(k // 16) for hh in range(2) for j in range(8)])
        w = w[:, perm]
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    planes = torch.stack([hi, lo], 0)                                   # (2, N, K)
    f = planes.view(2, n_pad // 32, 32, k // 16, 2, 8)                  # p, rt, i, ks, h, j
    f = f.permute(1, 3, 0, 4, 2, 5).contiguous()                        # rt, ks, p, h, i, j
    return f.view(n_pad // 32, k // 16, 2, 64, 8)


def pack_fragments16(weight2d, n_pad=None, acc_order=False):
    """(N, K) fp32 -> A-operand fragment order of v_mfma_f32_16x16x32_bf16 (cf_head_fused with mfma16):
    uint8 view of [N_pad/16][K/32][2 (hi, lo)][64 lanes][8 bf16]; lane (i = l & 15, g = l >> 4) holds
    W[16 rt + i][32 ks + 8 g + j], j = 0..7.
    acc_order (K = 256, the head output layer fed from accumulator registers): position (ks, g, j) holds hidden
    channel 64 (ks >> 1) + 16 (2 (ks & 1) + (j >> 2)) + 4 g + (j & 3) - the channels a lane of wave ks >> 1 finds
    in the two stacked 16 x 16 accumulators 2 (ks & 1), 2 (ks & 1) + 1."""
    n, k = weight2d.shape
    assert k % 32 == 0
    n_pad = n_pad or ((n + 15) // 16) * 16
    w = torch.zeros(n_pad, k)
    w[:n] = weight2d.float()
    if acc_order:
        assert k % 64 == 0
        perm = torch.tensor([64 * (ks >> 1) + 16 * (2 * (ks & 1) + (j >> 2)) + 4 * g + (j & 3)
                             for ks in range(k // 32) for g in range(4) for j in range(8)])
        w = w[:, perm]
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    planes = torch.stack([hi, lo], 0)                                   # (2, N, K)
    f = planes.view(2, n_pad // 16, 16, k // 32, 4, 8)                  # p, rt, i, ks, g, j
    f = f.permute(1, 3, 0, 4, 2, 5).contiguous()                        # rt, ks, p, g, i, j
    return f.view(n_pad // 16, k // 32, 2, 64, 8)


@dataclass
class PackedDcn:
    weight: torch.Tensor   # [N_pad, 9*C]  k = tap*C + c
    bias: torch.Tensor
    n: int
    n_pad: int
    c: int
    out_scale: float = 0.0   # cf_dcn_v2_f16x3: 2^-(s+4)

    def to(self, device):
        self.weight = self.weight.to(device).contiguous()
        self.bias = self.bias.to(device).contiguous()
        return self


def pack_dcn(weight, bias) -> PackedDcn:
    co, ci, kh, kw = weight.shape
    assert (kh, kw) == (3, 3) and ci % BK == 0
    n_pad = ((co + 31) // 32) * 32
    w = torch.zeros(n_pad, 9 * ci)
    w[:co] = weight.float().permute(0, 2, 3, 1).reshape(co, 9 * ci)
    b = torch.zeros(n_pad)
    b[:co] = bias
    return PackedDcn(w, b, co, n_pad, ci)


def pack_dcn_f16(weight, bias) -> PackedDcn:
    """cf_dcn_v2_f16x3: K order (tap, channel), weights 2^s-scaled, fp16 hi/lo, fragment order."""
    co, ci, kh, kw = weight.shape
    assert (kh, kw) == (3, 3) and ci % BK == 0
    n_pad = ((co + 31) // 32) * 32
    w = torch.zeros(n_pad, 9 * ci, dtype=torch.float64)
    w[:co] = weight.double().permute(0, 2, 3, 1).reshape(co, 9 * ci)
    wmax = float(w.abs().max())
    s_exp = int(torch.floor(torch.log2(torch.tensor(16384.0 / wmax)))) if wmax > 0 else 0
    ws = (w * 2.0 ** s_exp).float()
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    f = torch.stack([hi, lo], 0).view(2, n_pad // 32, 32, 9 * ci // 16, 2, 8).permute(1, 3, 0, 4, 2, 5).contiguous()
    b = torch.zeros(n_pad)
    b[:co] = bias
    pd = PackedDcn(f.view(n_pad // 32, 9 * ci // 16, 2, 64, 8), b, co, n_pad, ci)
    pd.out_scale = 2.0 ** -(s_exp + 4)
    return pd


def pack_upsample(weight):
    """ConvTranspose2d depthwise weight (C,1,k,k) -> [k][k][C]."""
    return weight.detach().float().cpu()[:, 0].permute(1, 2, 0).contiguous()


def _f16_split(w):
    """fp64/fp32 tensor -> (2^s * w) split into fp16 hi / lo, and the exponent s (max|w| -> [2^13, 2^14))."""
    wmax = float(w.abs().max())
    s_exp = int(torch.floor(torch.log2(torch.tensor(16384.0 / wmax)))) if wmax > 0 else 0
    ws = (w.double() * 2.0 ** s_exp).float()
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    return hi, lo, s_exp


@dataclass
class PackedStem:
    w_base: torch.Tensor
    b_base: torch.Tensor
    w_level0: torch.Tensor
    b_level0: torch.Tensor
    w_level1: torch.Tensor
    b_level1: torch.Tensor
    scale_base: float
    scale_level0: float
    scale_level1: float

    def to(self, device):
        for k in ("w_base", "b_base", "w_level0", "b_level0", "w_level1", "b_level1"):
            setattr(self, k, getattr(self, k).to(device).contiguous())
        return self


def pack_stem(w_base, b_base, w_l0, b_l0, w_l1, b_l1) -> PackedStem:
    """Weights of cf_stem_fused (BN already folded) in v_mfma_f32_16x16x32_f16 A-operand order: lane
    l = 16 kg + row holds the 8 k-values of k group kg.
      base_layer (16, C<=3, 7, 7): 13 k-steps of 4 taps (49 + 3 padding); a k group = ONE tap with
        k = [4 channels | the same 4 channels] - variant 0 holds {w_hi, w_hi}, variant 1 {w_lo, 0}, to meet
        the activations stored as [4 ch hi | 4 ch lo];
      level0 (16, 16, 3, 3) / level1 (32, 16, 3, 3): 5 k-steps of 2 taps (9 + 1 padding); k group kg =
        channels 8 (kg & 1) .. +8 of tap 2 ks + (kg >> 1); planes hi, lo; level1 has two 16-row tiles."""
    assert w_base.shape[0] == 16 and w_base.shape[1] <= 3 and tuple(w_base.shape[2:]) == (7, 7)
    assert tuple(w_l0.shape) == (16, 16, 3, 3) and tuple(w_l1.shape) == (32, 16, 3, 3)
    hi, lo, s0 = _f16_split(w_base)
    cb = w_base.shape[1]
    fb = torch.zeros(13, 2, 64, 8, dtype=torch.float16)
    for ks in range(13):
        for kg in range(4):
            tap = 4 * ks + kg
            if tap >= 49:
                continue
            ky, kx = divmod(tap, 7)
            rows = slice(16 * kg, 16 * kg + 16)
            fb[ks, 0, rows, 0:cb] = hi[:, :, ky, kx]
            fb[ks, 0, rows, 4:4 + cb] = hi[:, :, ky, kx]
            fb[ks, 1, rows, 0:cb] = lo[:, :, ky, kx]

    def frag3(w):
        h, l, s = _f16_split(w)
        n_rt = w.shape[0] // 16
        f = torch.zeros(n_rt, 5, 2, 64, 8, dtype=torch.float16)
        for rt in range(n_rt):
            for ks in range(5):
                for kg in range(4):
                    tap = 2 * ks + (kg >> 1)
                    if tap >= 9:
                        continue
                    ky, kx = divmod(tap, 3)
                    ch = slice(8 * (kg & 1), 8 * (kg & 1) + 8)
                    rows = slice(16 * kg, 16 * kg + 16)
                    f[rt, ks, 0, rows] = h[16 * rt:16 * rt + 16, ch, ky, kx]
                    f[rt, ks, 1, rows] = l[16 * rt:16 * rt + 16, ch, ky, kx]
        return f, s

    f0, s1 = frag3(w_l0)
    f1, s2 = frag3(w_l1)
    return PackedStem(fb, b_base.float().clone(), f0[0].contiguous(), b_l0.float().clone(), f1, b_l1.float().clone(),
                      2.0 ** -(s0 + 4), 2.0 ** -(s1 + 4), 2.0 ** -(s2 + 4))


# ------------------------------------------------------------------------------------------------
# Heads, first layer: "f16 main term + block-scaled FP6 cross terms" (cf_head_fused with mx = 1).
#
#   W * 2^s = Wh + Wl  (Wh fp16 RNE, Wl the exact remainder),   x * 16 = xh + xl  (the same, per element)
#   acc = Wh . xh                                  v_mfma_f32_16x16x32_f16, exact products
#       + q6(Wh) . q6(xl) + q6(Wl) . q6(xh)        ONE v_mfma_scale_f32_16x16x128_f8f6f4 per tap: K = [32 ch | 32 ch | 32 ch | 32 ch]
#   y   = acc * 2^-(s + 4) + b
#
# q6 = OCP MX FP6 e2m3 with one E8M0 scale per block of 32 channels of one tap (weights: per output row).  The scheme
# costs 1.5 MFMA passes per product instead of 3; its error (4 significant bits on terms that are 2^-11 of the product)
# passes the float64-anchored gate of tests/test_gpu_model.py ONLY on the first layer (docs/experiments/r5_heads_mx_numerics.txt),
# so the hidden and output layers stay bf16x3.
# ------------------------------------------------------------------------------------------------
MX_SLAB = 14592          # bytes of one (wave, tap) slab: 4 row tiles x 2 k-steps x 1 KiB | 4 x (1 KiB + 512 B) | 256 B of scales
MX_ROW = 272             # bytes of one pixel of the mx feature map: 64 fp16 | 4 x 32 B FP6 blocks | 4 scale bytes | pad


def e2m3_encode(t):
    """float64 tensor t (already divided by the block scale) -> uint8 e2m3 codes, RNE, saturating at 7.5; the sign bit
    follows signbit(t) also when the magnitude rounds to zero (as v_cvt_scalef32_*_fp6_* does)."""
    a = t.abs().clamp(max=7.5)
    q_lo = torch.round(a * 8.0)                  # [0, 2): step 1/8, codes 0..16
    q_mid = 16 + (torch.round(a * 4.0) - 8)      # [2, 4): step 1/4, codes 16..24
    q_hi = 24 + (torch.round(a * 2.0) - 8)       # [4, 7.5]: step 1/2, codes 24..31
    code = torch.where(a < 2.0, q_lo, torch.where(a < 4.0, q_mid, q_hi)).clamp(max=31).to(torch.int64)
    return (code | (torch.signbit(t).to(torch.int64) << 5)).to(torch.uint8)


def mx_block_exponent(amax):
    """smallest integer e with amax <= 7.5 * 2^e, from the fp32 bits of amax (the rule the kernels use): exponent field
    k, e = k - 2 + (mantissa > 0x700000).  amax == 0 -> -127 (scale byte 0)."""
    bits = amax.float().contiguous().view(torch.int32)
    k = ((bits >> 23) & 0xFF) - 127
    e = k - 2 + ((bits & 0x7FFFFF) > 0x700000).to(torch.int32)
    return torch.where(amax.float() == 0, torch.full_like(e, -127), e).clamp(-127, 127)


def pack_fp6_fields(codes):
    """(..., 32) uint8 codes -> (..., 24) uint8: element j in bits [6j, 6j + 6), little endian."""
    c = codes.to(torch.int64).reshape(*codes.shape[:-1], 8, 4)           # 4 codes -> 3 bytes
    w = c[..., 0] | (c[..., 1] << 6) | (c[..., 2] << 12) | (c[..., 3] << 18)
    out = torch.stack([w & 0xFF, (w >> 8) & 0xFF, (w >> 16) & 0xFF], -1)
    return out.reshape(*codes.shape[:-1], 24).to(torch.uint8)


def mx_quant_blocks(v):
    """v (..., 32 n) float -> (codes uint8 (..., n, 32), scale bytes uint8 (..., n)) with one E8M0 scale per 32 entries."""
    b = v.double().reshape(*v.shape[:-1], v.shape[-1] // 32, 32)
    e = mx_block_exponent(b.abs().amax(-1).float())
    codes = e2m3_encode(b / torch.pow(2.0, e.double()).unsqueeze(-1))
    return codes, (e + 127).to(torch.uint8)


def pack_head_first_mx(weight, bias, pc: bool, feat_scale: float = 16.0):
    """First 3x3 layer of one head (256, 64 [+3], 3, 3) -> the operand stream of head_patch16_kernel<..., MX>.

    feat_scale: the power of two the mx feature rows were written with (cf_pack_feat_mx: 16; a calibrated model may use less).
    Returns dict(w_first uint8, b_first f32 (256), first_scale 2^-s / feat_scale, real_cin).  Layout of w_first: for wave wv
    (64 output channels = 4 row tiles of 16) and tap t a slab of MX_SLAB bytes:
        [rt 4][ks 2][lane 64][8 fp16]   main term: lane (g = l >> 4, i = l & 15) holds Wh[64 wv + 16 rt + i][t, 32 ks + 8 g + j]
        [rt 4]([lane 64][16 B] | [lane 64][8 B])   cross term, 24 B per lane = 32 FP6 fields: g = 0, 1: q6(Wh) channels
                                        32 g .. +32 (to meet q6(xl)); g = 2, 3: q6(Wl) channels 32 (g - 2) .. +32 (to meet q6(xh))
        [lane 64][4 B]                  E8M0 scale bytes of that lane's block, byte rt
    then (pc only) the pc_hm part as bf16x3 fragments of W * 2^s * feat_scale: [wv 4][ks 3][rt 4][hi, lo][lane 64][8 bf16], k-step ks =
    taps 4 ks .. 4 ks + 3 x 8 channels (3 real)."""
    co, ci, kh, kw = weight.shape
    assert co == 256 and (kh, kw) == (3, 3) and ci == (67 if pc else 64)
    w = weight.double()
    wmax = float(w.abs().max())
    s_exp = int(torch.floor(torch.log2(torch.tensor(16384.0 / wmax)))) if wmax > 0 else 0
    ws = (w[:, :64] * 2.0 ** s_exp).float().permute(0, 2, 3, 1).reshape(256, 9, 64)      # (row, tap, channel)
    hi = ws.to(torch.float16)
    lo = ws - hi.float()                                                               # exact
    hc, hs = mx_quant_blocks(hi.float())                                                # (256, 9, 2, 32), (256, 9, 2)
    lc, ls = mx_quant_blocks(lo)
    hf, lf = pack_fp6_fields(hc), pack_fp6_fields(lc)                                   # (256, 9, 2, 24)
    out = torch.zeros(4, 9, MX_SLAB, dtype=torch.uint8)
    # main: [wv][tap][rt][ks][g][i][8] <- hi[(wv, rt, i)][tap][(ks, g, j)]
    m = hi.view(4, 4, 16, 9, 2, 4, 8).permute(0, 3, 1, 4, 5, 2, 6).contiguous()        # wv, tap, rt, ks, g, i, j
    out[:, :, :8192] = m.view(torch.uint8).reshape(4, 9, 8192)
    # cross: lane (g, i): g = 0, 1 -> hf block g; g = 2, 3 -> lf block g - 2
    x = torch.cat([hf, lf], 2).view(4, 4, 16, 9, 4, 24).permute(0, 3, 1, 4, 2, 5).contiguous()   # wv, tap, rt, g, i, 24
    x = x.view(4, 9, 4, 64, 24)
    out[:, :, 8192:8192 + 6144] = torch.cat([x[..., :16].reshape(4, 9, 4, 1024), x[..., 16:].reshape(4, 9, 4, 512)], -1).reshape(4, 9, 6144)
    sc = torch.cat([hs, ls], 2).view(4, 4, 16, 9, 4).permute(0, 3, 4, 2, 1).contiguous()          # wv, tap, g, i, rt
    out[:, :, 14336:] = sc.view(4, 9, 256)
    parts = [out.reshape(-1)]
    if pc:
        wp = torch.zeros(256, 12, 8, dtype=torch.float64)                               # (row, tap (9 real), 8 ch (3 real))
        wp[:, :9, :3] = (w[:, 64:67] * 2.0 ** s_exp * float(feat_scale)).permute(0, 2, 3, 1).reshape(256, 9, 3)
        wp = wp.float().view(256, 3, 32)                                                # k-step ks: k = 8 (tap - 4 ks) + c
        ph = wp.to(torch.bfloat16)
        pl = (wp - ph.float()).to(torch.bfloat16)
        f = torch.stack([ph, pl], 0).view(2, 4, 4, 16, 3, 4, 8).permute(1, 4, 2, 0, 5, 3, 6).contiguous()   # wv, ks, rt, plane, g, i, j
        parts.append(f.view(torch.uint8).reshape(-1))
    b = torch.zeros(256)
    b[:co] = bias.float()
    return dict(w_first=torch.cat(parts).contiguous(), b_first=b, first_scale=2.0 ** -s_exp / float(feat_scale),
                real_cin=(64, 3) if pc else (64,))
