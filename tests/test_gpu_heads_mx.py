"""GPU parity of the heads' first layer on "fp16 main term + block-scaled FP6 cross terms" (cf_head_fused with mx = 1,
cf_pack_feat_mx), through the C ABI, against oracle/mx_emul.py - an independent numpy / float64 restatement of the
quantiser, the bit packing and the arithmetic (reference layers: detectHeads.py:59-98, 165-191).

Bars: the feature rows are BYTES -> bit-exact.  The head outputs are floating point: against the oracle's own
evaluation of the SAME quantised operands (exact products, float64 accumulation) the kernel may differ by its fp32
accumulation and by the bf16x3 tail layers only: 2e-5 * max|ref| (+ 1e-5 per hidden layer); against plain fp32 torch
the scheme's own error shows: 2e-4 * max|ref| here (random weights; the end-to-end bound is test_gpu_model's)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import mx_emul


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def test_pack_feat_mx_rows_bit_exact(dev):
    """every byte of the 272-byte rows: fp16 plane, FP6 fields, E8M0 scale bytes, zero padding; blocks of zeros, values
    beyond the fp16 range, negative values, magnitudes over 30 binades, exact ties of the e2m3 grid"""
    from centerfusiondetect3d_amd import ops
    g = torch.Generator().manual_seed(3)
    M = 4099
    x = torch.randn(M, 64, generator=g) * torch.exp2(torch.randint(-20, 10, (M, 1), generator=g).float())
    x[5] = 0.0
    x[6, :32] = 0.0
    x[7] = 5000.0                                              # 16 x beyond 65504: clamped
    x[8] = -5000.0
    x[9] = torch.arange(64).float() * 0.0625 / 16              # on the grid and on its ties
    x[10, 1:] = 0.0
    x[11] = torch.relu(x[11])
    rows = ops.pack_feat_mx(x.view(1, M, 1, 64).to(dev)).view(M, 272).cpu().numpy()
    ref = mx_emul.feat_rows_ref(x.numpy())
    bad = np.nonzero((rows != ref).any(1))[0]
    assert bad.size == 0, (bad[:5], [np.nonzero(rows[b] != ref[b])[0][:8] for b in bad[:3]])
    # a wider tensor: only the first 64 channels are the feature map
    xw = torch.cat([x, torch.randn(M, 8, generator=g)], 1).contiguous()
    rows_w = ops.pack_feat_mx(xw.view(M, 72).to(dev)).cpu().numpy()
    assert np.array_equal(rows_w, ref)


def _heads_case(dev, n_hidden, radar, B, H, W, n_outs=(10, 1, 3, 8), acts=(2, 3, 0, 0), denormals=False):
    from centerfusiondetect3d_amd import ops, packing
    feat, pch = F.relu(rnd(B, 64, H, W, seed=1)) * 3.0, rnd(B, 3, H, W, seed=2) * 20.0
    feat[:, :, 0, :3] = 0.0                                    # all-zero pixels (zero blocks inside the image)
    if denormals:                                              # blocks whose maximum is an fp32 denormal / below 2^-125 (ADVICE r5:
        feat[:, :32, 1, :] = 1e-39                             #  their E8M0 code must stay a number - never 255 = NaN - and the
        feat[:, 32:, 2, :] = 2.0 ** -130                       #  head launch that reads them must stay finite)
        feat[:, :, 3, 1] = torch.tensor([1e-39, 0.0] * 32)
    rows = ops.pack_feat_mx(nhwc(feat).to(dev))
    srcs = [rows] + ([ops.split_bf16(nhwc(pch).to(dev), cs=8)] if radar else [])
    ci = 67 if radar else 64
    heads, refs64, refs32 = [], [], []
    for i, (no, act) in enumerate(zip(n_outs, acts)):
        w1, b1 = rnd(256, ci, 3, 3, seed=300 + i, scale=(ci * 9) ** -0.5), rnd(256, seed=310 + i, scale=0.1)
        x64 = torch.relu(mx_emul.first_layer_mx(feat, pch if radar else None, w1, b1))
        x32 = F.relu(F.conv2d(torch.cat([feat, pch], 1) if radar else feat, w1, b1, 1, 1))
        first = packing.pack_head_first_mx(w1, b1, radar)
        wh, bh = [], []
        for l in range(n_hidden):
            w, b = rnd(256, 256, 1, 1, seed=10 * i + l, scale=1 / 16), rnd(256, seed=50 + 10 * i + l, scale=0.1)
            x64 = torch.relu(F.conv2d(x64, w.double(), b.double()))
            x32 = F.relu(F.conv2d(x32, w, b))
            wh.append(packing.pack_fragments16(w.view(256, 256)).to(dev)); bh.append(b.to(dev))
        w, b = rnd(no, 256, 1, 1, seed=100 + i, scale=1 / 16), rnd(no, seed=200 + i)
        refs64.append(F.conv2d(x64, w.double(), b.double()))
        refs32.append(F.conv2d(x32, w, b))
        b32 = torch.zeros(32); b32[:no] = b
        out = torch.full((B, no, H, W), float("nan"), device=dev)
        out2 = torch.full((B, no, H, W), float("nan"), device=dev) if act == 3 else None
        heads.append(dict(w_first=first["w_first"].to(dev), b_first=first["b_first"].to(dev), first_scale=first["first_scale"],
                          w_hidden=wh, b_hidden=bh, w_out=packing.pack_fragments16(w.view(no, 256)).to(dev), b_out=b32.to(dev),
                          w_out_perm=packing.pack_fragments16(w.view(no, 256), acc_order=True).to(dev),
                          mfma16=True, n_out=no, act=act, out=out, out2=out2))
    f = ops.head_fused_args(srcs, [64, 8][:len(srcs)], None, 0, B, H, W, heads)
    assert f.mx == 1 and f.mfma16 == 1 and f.layout3x3 == 1
    f._keep = srcs                                             # the argument block holds raw pointers only
    return f, heads, refs64, refs32


@pytest.mark.parametrize("n_hidden,radar,B,H,W", [
    (0, False, 2, 9, 14),        # one partial tile per image
    (0, True, 1, 21, 37),        # pc_hm taps, ragged tiles in both directions
    (0, False, 3, 16, 32),       # exact tiling, several heads per workgroup
    (2, True, 2, 13, 19),        # hidden layers behind the mx first layer (two 64-pixel halves)
    (1, True, 1, 8, 40),
])
def test_head_fused_mx_whole_head(dev, n_hidden, radar, B, H, W):
    import os
    from centerfusiondetect3d_amd import ops
    f, heads, refs64, refs32 = _heads_case(dev, n_hidden, radar, B, H, W)
    firsts = None
    for tile in ("0", "1"):                                    # both tile orientations: identical bits
        os.environ["CF_HEAD_TILE"] = tile
        try:
            for hd in heads:
                hd["out"].fill_(float("nan"))
            ops.run_head_fused(f)
        finally:
            del os.environ["CF_HEAD_TILE"]
        outs = [hd["out"].clone() for hd in heads] + [hd["out2"].clone() for hd in heads if hd["out2"] is not None]
        if firsts is None:
            firsts = outs
        else:
            assert all(torch.equal(a, b) for a, b in zip(firsts, outs))
    ops.run_head_fused(f)
    for hd, r64, r32 in zip(heads, refs64, refs32):
        scale = float(r64.abs().max())
        got = hd["out"].cpu().double()
        assert bool(torch.isfinite(got).all())
        if hd["act"] == 2:
            torch.testing.assert_close(got, torch.clamp(torch.sigmoid(r64), 1e-4, 1 - 1e-4), rtol=1e-4, atol=1e-5)
            continue
        e_emul = float((got - r64).abs().max()) / scale
        e_fp32 = float((got - r32.double()).abs().max()) / scale
        print(f"[mx] n_out {hd['n_out']:2d}: vs oracle mx arithmetic {e_emul:.2e}, vs fp32 torch {e_fp32:.2e}")
        assert e_emul < 2e-5 + 1e-5 * n_hidden, e_emul
        assert e_fp32 < 2e-4, e_fp32
        if hd["act"] == 3:
            torch.testing.assert_close(hd["out2"].cpu().double(), 1.0 / (torch.sigmoid(r64) + 1e-6) - 1.0, rtol=1e-3, atol=1e-3)


def test_head_fused_mx_is_finite_on_denormal_blocks(dev):
    """Feature blocks whose largest value is an fp32 denormal (or below 2^-125, where the block exponent would leave E8M0's
    range): the pack kernel writes them as zero blocks with a valid scale byte, and the head launch that reads them stays
    finite and equal to the oracle's evaluation of the same operands."""
    from centerfusiondetect3d_amd import ops
    f, heads, refs64, _ = _heads_case(dev, 0, True, 1, 8, 16, denormals=True)
    rows = f._keep[0].cpu().numpy().reshape(-1, 272)
    assert rows[:, 256:260].max() < 255
    ops.run_head_fused(f)
    for hd, r64 in zip(heads, refs64):
        got = hd["out"].cpu().double()
        assert bool(torch.isfinite(got).all())
        if hd["act"] != 2:
            assert float((got - r64).abs().max()) / float(r64.abs().max()) < 2e-5


def test_head_fused_mx_refuses_what_it_cannot_read(dev):
    """mx streams are readable by the 16x16x32 patch kernel only: without mfma16 fragments for the tail layers the launch
    is refused, not run"""
    from centerfusiondetect3d_amd import ops, _lib
    f, heads, _, _ = _heads_case(dev, 0, False, 1, 8, 16, n_outs=(3,), acts=(0,))
    ops.run_head_fused(f)
    f.mfma16 = 0
    with pytest.raises(_lib.CfHipError, match="mx"):
        ops.run_head_fused(f)
    # hidden layers behind an mx first layer exist with the pc_hm source only (the radar heads); without it: refused
    f, heads, _, _ = _heads_case(dev, 1, False, 1, 8, 16, n_outs=(3,), acts=(0,))
    with pytest.raises(_lib.CfHipError, match="pc_hm"):
        ops.run_head_fused(f)


def test_dcn_epilogue_writes_the_same_mx_rows(dev):
    """cf_dcn_v2_f16x3 with out_mx: the rows the DCN's epilogue writes are, byte for byte, what cf_pack_feat_mx makes of the
    layer's fp32 output (and so the oracle's rows of it) - on both tile sizes of the 64-channel kernel, ragged last tile"""
    import os
    from centerfusiondetect3d_amd import ops, packing
    for (B, H, W) in ((2, 23, 37), (1, 8, 16)):
        g = torch.Generator().manual_seed(B)
        x = torch.randn(B, H, W, 64, generator=g).to(dev)
        om = torch.zeros(B, H, W, 32)
        om[..., :18] = torch.randn(B, H, W, 18, generator=g) * 1.5
        om[..., 18:27] = torch.randn(B, H, W, 9, generator=g)
        om = om.to(dev)
        pd = packing.pack_dcn_f16(torch.randn(64, 64, 3, 3, generator=g) * 0.05, torch.randn(64, generator=g) * 0.1).to(dev)
        out = torch.empty(B, H, W, 64, device=dev)
        rows = torch.zeros(B, H, W, 272, device=dev, dtype=torch.uint8)
        a = ops.dcn_args(pd, x, om, 32, B, H, W, out, 64, out_mx=rows)
        ops.run_dcn(a)
        ref_rows = ops.pack_feat_mx(out)
        assert torch.equal(rows, ref_rows)
        assert np.array_equal(rows.view(-1, 272).cpu().numpy(), mx_emul.feat_rows_ref(out.view(-1, 64).cpu().numpy()))
        assert float(out.abs().max()) > 0 and bool((out >= 0).all())            # ReLU applied, not a dead layer
