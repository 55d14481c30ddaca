"""GPU: the operator-level drop-in `centerfusiondetect3d_amd.ops.deform_conv2d` - torchvision's signature and
semantics (NCHW, activated mask, raw weight / bias) as the reference's `DeformConv.forward` calls it
(/root/reference/src/lib/model/networks/dla.py:456-472) - against the CPU oracle
`oracle/dcn_ref.deform_conv2d` on the six DeformConv channel pairs of SURVEY Appendix A, and through the eight
known-answer tests that pin the oracle itself (tests/test_oracle_dcn.py), here run through the HIP kernel.
Tolerance: max|err| <= 5e-6 max|ref| against the float64 oracle (fp32-level: split-fp16 products, fp32 sums)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import dcn_ref

S, P, D = (1, 1), (1, 1), (1, 1)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _op(dev, x, off, w, b, mask, **kw):
    from centerfusiondetect3d_amd import ops
    d = lambda t: None if t is None else t.to(dev)
    kw = dict(dict(stride=S, padding=P, dilation=D), **kw)
    return ops.deform_conv2d(input=d(x), offset=d(off), weight=d(w), bias=d(b), mask=d(mask), **kw).cpu()


def _relerr(got, ref):
    return float((got.double() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-30))


# the six (Cin, Cout, map) pairs of the 16 DeformConv layers, maps shrunk where the CPU oracle
# would take minutes, plus the full 64 -> 64 node layer size once
@pytest.mark.parametrize("B,Ci,Co,H,W,mag", [(1, 512, 256, 14, 25, 1.0), (2, 256, 256, 28, 50, 2.0),
                                             (1, 256, 128, 28, 50, 3.0), (1, 128, 128, 56, 100, 2.0),
                                             (2, 128, 64, 56, 100, 8.0), (1, 64, 64, 112, 200, 2.0),
                                             (1, 256, 64, 28, 50, 30.0)])
def test_operator_matches_oracle_on_the_reference_channel_pairs(dev, B, Ci, Co, H, W, mag):
    x, off = rnd(B, Ci, H, W, seed=1), rnd(B, 18, H, W, seed=2, scale=mag)
    mask = torch.sigmoid(rnd(B, 9, H, W, seed=3))
    w, b = rnd(Co, Ci, 3, 3, seed=4, scale=(Ci * 9) ** -0.5), rnd(Co, seed=5)
    ref = dcn_ref.deform_conv2d(x.double(), off.double(), w.double(), b.double(), S, P, D, mask.double())
    got = _op(dev, x, off, w, b, mask)
    assert got.shape == ref.shape == (B, Co, H, W) and got.dtype == torch.float32
    err = _relerr(got, ref)
    print(f"[deform_conv2d] {Ci}->{Co} @{H}x{W}: max|err|/max|ref| = {err:.2e}")
    assert err < 5e-6, err


def test_the_reference_deformconv_call_sequence_runs_on_the_operator(dev):
    """DeformConv.forward as the reference spells it (dla.py:456-472): conv_offset_mask -> chunk(3) -> cat(o1, o2) ->
    sigmoid(mask) -> deform_conv2d(input=, offset=, weight=, bias=, stride=, padding=, dilation=, mask=) -> BN -> ReLU,
    with `deform_conv2d` bound to the drop-in (what `torchvision.ops.deform_conv2d = ops.deform_conv2d` does)."""
    from centerfusiondetect3d_amd.ops import deform_conv2d
    B, Ci, Co, H, W = 2, 64, 64, 40, 56
    x = rnd(B, Ci, H, W, seed=1)
    com_w, com_b = rnd(27, Ci, 3, 3, seed=2, scale=0.02), rnd(27, seed=3)
    w, b = rnd(Co, Ci, 3, 3, seed=4, scale=1 / 24), rnd(Co, seed=5)
    bn = (torch.rand(Co) + 0.5, rnd(Co, seed=6, scale=0.1), rnd(Co, seed=7, scale=0.1), torch.rand(Co) + 0.5)

    def forward(x, op, dv):
        t = lambda v: v.to(dv)
        offset_mask = F.conv2d(x, t(com_w), t(com_b), stride=(1, 1), padding=(1, 1))
        offset1, offset2, mask = torch.chunk(offset_mask, 3, dim=1)
        offset = torch.cat((offset1, offset2), dim=1)
        mask = torch.sigmoid(mask)
        y = op(input=x, offset=offset, weight=t(w), bias=t(b), stride=(1, 1), padding=(1, 1), dilation=(1, 1), mask=mask)
        return F.relu(F.batch_norm(y, t(bn[2]), t(bn[3]), t(bn[0]), t(bn[1]), False, 0.1, 1e-5))

    ref = forward(x, dcn_ref.deform_conv2d, "cpu")
    got = forward(x.to(dev), deform_conv2d, dev).cpu()
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))


def test_weights_are_packed_once_per_version(dev):
    from centerfusiondetect3d_amd import ops
    x, off = rnd(1, 32, 9, 11, seed=1).to(dev), rnd(1, 18, 9, 11, seed=2).to(dev)
    w = torch.nn.Parameter(rnd(32, 32, 3, 3, seed=3).to(dev))
    ops._DCN_PACKS.clear()
    a = ops.deform_conv2d(x, off, w, None, S, P, D, None)
    assert len(ops._DCN_PACKS) == 1
    assert torch.equal(a, ops.deform_conv2d(x, off, w, None, S, P, D, None)) and len(ops._DCN_PACKS) == 1
    with torch.no_grad():
        w.mul_(2.0)                                        # in-place update bumps the version: re-packed
    b = ops.deform_conv2d(x, off, w, None, S, P, D, None)
    assert len(ops._DCN_PACKS) == 1
    torch.testing.assert_close(b, 2 * a, rtol=1e-5, atol=1e-5)
    # a DIFFERENT tensor that lands on the freed one's address (and id) must not hit the old entry
    outs = []
    for k in range(6):
        w2 = rnd(32, 32, 3, 3, seed=10 + k).to(dev)
        outs.append((ops.deform_conv2d(x, off, w2, None, S, P, D, None).cpu(), w2.cpu()))
        del w2
    from oracle import dcn_ref as _ref
    for got, wk in outs:
        torch.testing.assert_close(got, _ref.deform_conv2d(x.cpu(), off.cpu(), wk, None, S, P, D, None), rtol=1e-4, atol=1e-4)


def test_writes_through_dot_data_are_seen(dev):
    """`w.data.copy_()` / `w.data.mul_()` / `w.data = ...` (an EMA swap, a hand-rolled checkpoint load) bump no version
    counter of the Parameter: the packed-weight cache notices through the data pointer and the checksum of the live bits;
    with verification off, `clear_dcn_pack_cache()` is the explicit way."""
    from centerfusiondetect3d_amd import ops
    x, off = rnd(1, 32, 9, 11, seed=1).to(dev), rnd(1, 18, 9, 11, seed=2).to(dev)
    w = torch.nn.Parameter(rnd(32, 32, 3, 3, seed=3).to(dev))
    bias = torch.nn.Parameter(rnd(32, seed=4).to(dev))
    ops.clear_dcn_pack_cache()
    a = ops.deform_conv2d(x, off, w, bias, S, P, D, None)
    v0 = w._version
    w.data.mul_(2.0)
    bias.data.mul_(2.0)
    assert w._version == v0                                  # (what makes this case invisible to a version check)
    torch.testing.assert_close(ops.deform_conv2d(x, off, w, bias, S, P, D, None), 2 * a, rtol=1e-5, atol=1e-5)
    w.data.copy_(w.data * 0.5)
    bias.data.copy_(bias.data * 0.5)
    torch.testing.assert_close(ops.deform_conv2d(x, off, w, bias, S, P, D, None), a, rtol=1e-5, atol=1e-5)
    w.data = (w.data * 4.0).clone()                          # a new storage behind the same Parameter object
    bias.data = (bias.data * 4.0).clone()                    # (powers of two: the packed split is the same up to the exponent)
    torch.testing.assert_close(ops.deform_conv2d(x, off, w, bias, S, P, D, None), 4 * a, rtol=1e-5, atol=1e-5)
    assert len(ops._DCN_PACKS) == 1
    prev = ops.set_dcn_pack_verify(False)
    try:
        assert prev is True
        b3 = ops.deform_conv2d(x, off, w, bias, S, P, D, None)
        w.data.mul_(2.0)
        bias.data.mul_(2.0)
        assert torch.equal(ops.deform_conv2d(x, off, w, bias, S, P, D, None), b3)     # frozen-weights mode: stale by contract
        ops.clear_dcn_pack_cache()
        torch.testing.assert_close(ops.deform_conv2d(x, off, w, bias, S, P, D, None), 8 * a, rtol=1e-5, atol=1e-5)
    finally:
        ops.set_dcn_pack_verify(True)


# ---- the known-answer tests of tests/test_oracle_dcn.py, through the HIP kernel (Cin padded to the kernel's 32) ----
def test_kat_zero_offset_unit_mask_is_conv2d(dev):
    x, w, b = rnd(2, 32, 13, 17), rnd(6, 32, 3, 3, seed=1, scale=1 / 17), rnd(6, seed=2)
    got = _op(dev, x, torch.zeros(2, 18, 13, 17), w, b, torch.ones(2, 9, 13, 17))
    torch.testing.assert_close(got, F.conv2d(x, w, b, 1, 1), rtol=1e-5, atol=1e-5)


def test_kat_mask_none_is_unmodulated(dev):
    x, w = rnd(1, 32, 8, 9), rnd(5, 32, 3, 3, seed=1, scale=1 / 17)
    off = rnd(1, 18, 8, 9, seed=2)
    torch.testing.assert_close(_op(dev, x, off, w, None, None), _op(dev, x, off, w, None, torch.ones(1, 9, 8, 9)),
                               rtol=0, atol=0)
    torch.testing.assert_close(_op(dev, x, off, w, None, None), dcn_ref.deform_conv2d(x, off, w, None, S, P, D, None),
                               rtol=1e-5, atol=1e-5)


def test_kat_integer_offset_is_shifted_conv(dev):
    x, w = rnd(1, 32, 12, 15), rnd(3, 32, 3, 3, seed=1, scale=1 / 17)
    dy, dx = 2, -3
    off = torch.zeros(1, 18, 12, 15)
    off[:, 0::2] = dy
    off[:, 1::2] = dx
    Pd = 5
    full = F.conv2d(F.pad(x, (Pd, Pd, Pd, Pd)), w)
    exp = full[:, :, Pd - 1 + dy:Pd - 1 + dy + 12, Pd - 1 + dx:Pd - 1 + dx + 15]
    torch.testing.assert_close(_op(dev, x, off, w, None, None), exp, rtol=1e-5, atol=1e-5)


def test_kat_offset_channel_order_dy_then_dx_per_tap(dev):
    x, w = rnd(1, 32, 9, 9), torch.zeros(1, 32, 3, 3)
    w[0, :, 1, 2] = 1.0
    off = torch.zeros(1, 18, 9, 9)
    off[:, 2 * 5] = 1.0                              # tap k = 5 (i=1, j=2) one row down
    exp = torch.zeros(1, 1, 9, 9)
    exp[:, 0, :8, :8] = x[:, :, 1:, 1:].sum(1)
    torch.testing.assert_close(_op(dev, x, off, w, None, None), exp, rtol=1e-5, atol=1e-5)


def test_kat_mask_is_linear_per_tap(dev):
    x, w = rnd(1, 32, 8, 8), rnd(2, 32, 3, 3, seed=1, scale=1 / 17)
    off = rnd(1, 18, 8, 8, seed=2)
    g = torch.Generator().manual_seed(3)
    m1, m2 = torch.rand(1, 9, 8, 8, generator=g), torch.rand(1, 9, 8, 8, generator=g)
    f = lambda m: _op(dev, x, off, w, None, m)
    torch.testing.assert_close(f(m1 + 2 * m2), f(m1) + 2 * f(m2), rtol=1e-4, atol=1e-5)


def test_kat_all_out_of_range_gives_bias(dev):
    x, w, b = rnd(1, 32, 6, 6), rnd(4, 32, 3, 3, seed=1), rnd(4, seed=2)
    got = _op(dev, x, torch.full((1, 18, 6, 6), 100.0), w, b, torch.ones(1, 9, 6, 6))
    torch.testing.assert_close(got, b.view(1, 4, 1, 1).expand(1, 4, 6, 6).contiguous(), rtol=0, atol=1e-6)


def test_kat_half_pixel_is_mean_of_integer_neighbours(dev):
    x, w = rnd(1, 32, 10, 10), rnd(2, 32, 3, 3, seed=1, scale=1 / 17)

    def run(dx):
        off = torch.zeros(1, 18, 10, 10)
        off[:, 1::2] = dx
        return _op(dev, x, off, w, None, None)
    torch.testing.assert_close(run(0.5), 0.5 * (run(0.0) + run(1.0)), rtol=1e-5, atol=1e-5)


def test_kat_border_rule_minus_one_exclusive(dev):
    x = torch.ones(1, 32, 4, 4)
    w = torch.zeros(1, 32, 3, 3)
    w[0, 0, 1, 1] = 1.0
    off = torch.zeros(1, 18, 4, 4)
    off[:, 8] = -0.25
    out = _op(dev, x, off, w, None, None)
    assert torch.allclose(out[0, 0, 0], torch.full((4,), 0.75)) and torch.allclose(out[0, 0, 1:], torch.ones(3, 4))
    off[:, 8] = -1.0
    out = _op(dev, x, off, w, None, None)
    assert torch.all(out[0, 0, 0] == 0) and torch.all(out[0, 0, 1:] == 1)


def test_unsupported_configurations_raise_instead_of_falling_back(dev):
    from centerfusiondetect3d_amd import ops, _lib
    x, off, w = rnd(1, 32, 8, 8).to(dev), torch.zeros(1, 18, 8, 8, device=dev), rnd(4, 32, 3, 3).to(dev)
    with pytest.raises(NotImplementedError):
        ops.deform_conv2d(x, off, w, None, stride=(2, 2), padding=P, dilation=D)
    with pytest.raises(NotImplementedError):
        ops.deform_conv2d(x, off, w)                                # torchvision's default padding is 0
    with pytest.raises(NotImplementedError):
        ops.deform_conv2d(x, off, rnd(4, 16, 3, 3).to(dev), None, S, P, D)   # groups = 2
    with pytest.raises(NotImplementedError):
        ops.deform_conv2d(rnd(1, 24, 8, 8).to(dev), off, rnd(4, 24, 3, 3).to(dev), None, S, P, D)
    with pytest.raises(ValueError):
        ops.deform_conv2d(x, torch.zeros(1, 18, 7, 8, device=dev), w, None, S, P, D)
    with pytest.raises(_lib.CfHipError):
        ops.deform_conv2d(x.cpu(), off.cpu(), w.cpu(), None, S, P, D)


@pytest.mark.parametrize("C,H,W,S_,o", [(18, 9, 13, 32, 0), (9, 9, 13, 32, 18), (64, 33, 35, 64, 0), (3, 5, 7, 8, 4)])
def test_nchw_to_nhwc(dev, C, H, W, S_, o):
    from centerfusiondetect3d_amd import ops
    x = rnd(2, C, H, W, seed=1)
    out = torch.full((2, H, W, S_), -7.0, device=dev)
    ops.nchw_to_nhwc(x.to(dev), out, o)
    exp = torch.full((2, H, W, S_), -7.0)
    exp[..., o:o + C] = x.permute(0, 2, 3, 1)
    assert torch.equal(out.cpu(), exp)
