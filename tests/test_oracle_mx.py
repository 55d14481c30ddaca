"""CPU: the oracle restatement of the heads' first-layer arithmetic (oracle/mx_emul.py: FP6 e2m3 quantiser, E8M0 block
exponents, bit packing, 272-byte feature rows) against known answers, against the product's independent host-side packer
(packing.py, torch) and - by decoding the operand stream packing.pack_head_first_mx writes exactly as the kernel's lanes
address it - against its own convolution.  Reference layers: detectHeads.py:59-79, 165-191.  No GPU."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import mx_emul
from centerfusiondetect3d_amd import packing


def test_e2m3_grid_known_answers():
    vals = mx_emul.e2m3_values(np.arange(64, dtype=np.uint8))
    assert vals[:8].tolist() == [0, .125, .25, .375, .5, .625, .75, .875]           # subnormals: m / 8
    assert vals[8:16].tolist() == [1, 1.125, 1.25, 1.375, 1.5, 1.625, 1.75, 1.875]
    assert vals[16] == 2.0 and vals[24] == 4.0 and vals[31] == 7.5 and vals[63] == -7.5 and vals[32] == 0.0
    t = np.array([0.0, -0.03, 0.0625, 0.1875, 1.9375, 2.1, 3.9, 4.25, 7.3, 9.0, -7.75])
    # nearest, ties to the even code, saturating, sign kept on a zero result
    assert mx_emul.e2m3_codes(t).tolist() == [0, 32, 0, 2, 16, 16, 24, 24, 31, 31, 63]
    assert packing.e2m3_encode(torch.from_numpy(t)).tolist() == mx_emul.e2m3_codes(t).tolist()
    e = mx_emul.block_exponent(np.array([0.0, 7.5, 7.50001, 1.0, 1.875, 1.9, 3.75, 4.0], np.float32))
    assert e.tolist() == [-127, 0, 1, -2, -2, -1, -1, 0]                              # smallest e with amax <= 7.5 * 2^e
    assert packing.mx_block_exponent(torch.tensor([0.0, 7.5, 7.50001, 1.0, 1.875, 1.9, 3.75, 4.0])).tolist() == e.tolist()


def test_quantiser_and_bit_packing_agree_with_the_host_packer():
    g = torch.Generator().manual_seed(0)
    v = torch.randn(50, 96, generator=g) * torch.exp2(torch.randint(-12, 6, (50, 1), generator=g).float())
    v[3] = 0.0
    c0, e0, dq = mx_emul.quant_blocks(v.numpy())
    c1, s1 = packing.mx_quant_blocks(v)
    assert np.array_equal(c0, c1.numpy()) and np.array_equal(e0 + 127, s1.numpy().astype(np.int32))
    assert np.array_equal(mx_emul.pack_fields(c0), packing.pack_fp6_fields(c1).numpy())
    # error at most half a step of the top binade: 0.25 * 2^e with amax > 3.75 * 2^e
    amax = np.abs(v.numpy().reshape(50, 3, 32)).max(-1, keepdims=True)
    assert (np.abs(dq.reshape(50, 3, 32) - v.numpy().reshape(50, 3, 32)) <= amax / 15.0 + 1e-30).all()
    # field j sits in bits [6j, 6j + 6)
    one = np.zeros((1, 32), np.uint8); one[0, 5] = 63
    b = mx_emul.pack_fields(one)[0]
    assert int.from_bytes(bytes(b), "little") == 63 << 30


def test_feature_rows_layout():
    x = np.abs(np.random.RandomState(1).standard_normal((7, 64)).astype(np.float32)) * 2
    rows = mx_emul.feat_rows_ref(x)
    assert rows.shape == (7, mx_emul.ROW) and mx_emul.ROW == packing.MX_ROW == 272
    hi, ql, qh = mx_emul.rows_unpack(rows)
    h_ref, l_ref = mx_emul.split_f16(x * np.float32(16))
    assert np.array_equal(hi, h_ref.astype(np.float64))
    assert np.array_equal(qh, mx_emul.quant_blocks(h_ref)[2]) and np.array_equal(ql, mx_emul.quant_blocks(l_ref)[2])
    # segment g of a row: channels 8g.. and 32 + 8g.. as fp16, then FP6 block g; padding is zero
    assert np.array_equal(rows[:, 64:80].copy().view(np.float16).astype(np.float32), h_ref[:, 8:16])
    assert np.array_equal(rows[:, 80:96].copy().view(np.float16).astype(np.float32), h_ref[:, 40:48])
    for g in range(4):
        assert not rows[:, 64 * g + 56:64 * g + 64].any()
    assert not rows[:, 260:].any()


def _decode_stream(w_first, pc):
    """the operand stream as the kernel's lanes address it -> (Wh (256, 9, 64) fp16 values, q6(Wh), q6(Wl) dequantised,
    pc part (256, 12, 8) or None), all float64"""
    S = packing.MX_SLAB
    raw = w_first.numpy()
    wh = np.zeros((256, 9, 64)); wh6 = np.zeros((256, 9, 64)); wl6 = np.zeros((256, 9, 64))
    for wv in range(4):
        for tap in range(9):
            slab = raw[(wv * 9 + tap) * S:(wv * 9 + tap + 1) * S]
            for rt in range(4):
                for lane in range(64):
                    g, i = lane >> 4, lane & 15
                    row = 64 * wv + 16 * rt + i
                    for ks in range(2):
                        o = (rt * 2 + ks) * 1024 + lane * 16
                        wh[row, tap, 32 * ks + 8 * g:32 * ks + 8 * g + 8] = slab[o:o + 16].copy().view(np.float16)
                    f = np.concatenate([slab[8192 + rt * 1536 + lane * 16:][:16], slab[8192 + rt * 1536 + 1024 + lane * 8:][:8]])
                    bits = int.from_bytes(bytes(f), "little")
                    codes = np.array([(bits >> (6 * j)) & 63 for j in range(32)], np.uint8)
                    sc = np.ldexp(1.0, int(slab[14336 + lane * 4 + rt]) - 127)
                    dst = wh6 if g < 2 else wl6
                    dst[row, tap, 32 * (g & 1):32 * (g & 1) + 32] = mx_emul.e2m3_values(codes) * sc
    wp = None
    if pc:
        base = 4 * 9 * S
        wp = np.zeros((256, 12, 8))
        bf = lambda b: (b.copy().view(np.uint16).astype(np.uint32) << 16).view(np.float32).astype(np.float64)
        for wv in range(4):
            for ks in range(3):
                for rt in range(4):
                    for lane in range(64):
                        g, i = lane >> 4, lane & 15
                        o = base + ((wv * 3 + ks) * 4 + rt) * 2048 + lane * 16
                        wp[64 * wv + 16 * rt + i, 4 * ks + g] = bf(raw[o:o + 16]) + bf(raw[o + 1024:o + 1024 + 16])
    return wh, wh6, wl6, wp


def test_operand_stream_decodes_to_the_oracle_arithmetic():
    """pack_head_first_mx's bytes, read back lane by lane, give the same first layer as oracle.mx_emul.first_layer_mx"""
    g = torch.Generator().manual_seed(5)
    w = torch.randn(256, 67, 3, 3, generator=g) * 0.04
    b = torch.randn(256, generator=g) * 0.1
    feat = F.relu(torch.randn(1, 64, 5, 6, generator=g)) * 2
    pch = torch.randn(1, 3, 5, 6, generator=g) * 10
    d = packing.pack_head_first_mx(w, b, True)
    assert d["w_first"].numel() == 4 * 9 * packing.MX_SLAB + 4 * 3 * 4 * 2048
    wh, wh6, wl6, wp = _decode_stream(d["w_first"], True)
    rows = mx_emul.feat_rows_ref(feat.permute(0, 2, 3, 1).reshape(-1, 64).numpy())
    xh, xl6, xh6 = (a.reshape(5, 6, 64) for a in mx_emul.rows_unpack(rows))
    pad = lambda a: np.pad(a, ((1, 1), (1, 1), (0, 0)))
    xhp, xl6p, xh6p = pad(xh), pad(xl6), pad(xh6)
    ph = pch[0].permute(1, 2, 0).to(torch.bfloat16).double().numpy()
    pl = (pch[0].permute(1, 2, 0).double() - torch.from_numpy(ph)).float().to(torch.bfloat16).double().numpy()
    pcp = np.pad(np.concatenate([ph + pl, np.zeros((5, 6, 5))], -1), ((1, 1), (1, 1), (0, 0)))   # hi + lo: both planes meet hi + lo weights up to lo.lo
    y, yp = np.zeros((256, 5, 6)), np.zeros((256, 5, 6))
    for tap in range(9):
        dy, dx = divmod(tap, 3)
        sl = (slice(dy, dy + 5), slice(dx, dx + 6))
        y += np.einsum("nc,hwc->nhw", wh[:, tap], xhp[sl]) + np.einsum("nc,hwc->nhw", wh6[:, tap], xl6p[sl]) \
            + np.einsum("nc,hwc->nhw", wl6[:, tap], xh6p[sl])
        yp += np.einsum("nc,hwc->nhw", wp[:, tap], pcp[sl])
    bias = d["b_first"].double().numpy()[:, None, None]
    # the feature part is EXACTLY the oracle's arithmetic on the same quantised operands
    ref0 = mx_emul.first_layer_mx(feat, torch.zeros_like(pch), w, b)[0].numpy()
    assert np.abs(y * d["first_scale"] + bias - ref0).max() <= 1e-12 * np.abs(ref0).max()
    # the pc_hm part: the decoded (hi + lo) weights times (hi + lo) activations carry the lo.lo product bf16x3 drops
    ref = mx_emul.first_layer_mx(feat, pch, w, b)[0].numpy()
    assert np.abs((y + yp) * d["first_scale"] + bias - ref).max() <= 1e-5 * np.abs(ref).max()
    # and the scheme itself against float64: a few 1e-5 of the output range
    r64 = F.conv2d(torch.cat([feat, pch], 1).double(), w.double(), b.double(), 1, 1)[0].numpy()
    assert np.abs(ref - r64).max() <= 5e-5 * np.abs(r64).max()
