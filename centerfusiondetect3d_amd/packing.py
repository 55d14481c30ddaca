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
    scale = g / torch.sqrt(var + BN_EPS)
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


def pack_conv_f16(weight, bias, sources: Sequence[Source], stride=1, pad=None, dilation=1) -> PackedConv:
    """Packing for cf_conv2d_f16x3: fp32 NHWC sources, 8-channel slots (4 per 32-deep chunk), weights
    scaled by 2^s (max|w| -> [2^13, 2^14)), split into fp16 hi/lo and laid out in MFMA A-operand
    fragment order.  PackedConv.out_scale = 2^-(s+4) undoes the weight and activation scales.

    A 3x3 / pad 1 convolution (stride 1 or 2) of ONE source with C % 16 == 0 is packed SLICE-MAJOR
    (k = (16-channel slice, tap, channel)): that is the order cf_conv3x3_f16x3 (LDS patch reuse)
    consumes, and since the slot table spells the same order out the generic kernel runs the very same
    weights (PackedConv.patch marks them)."""
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
    k_pad = len(slots) * 8
    w = torch.zeros(n_pad, k_pad, dtype=torch.float64)
    wf = weight.double()
    for j, (c0, r, q) in enumerate(cols):
        if c0 >= 0:
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
