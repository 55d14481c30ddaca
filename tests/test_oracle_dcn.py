"""CPU: known-answer tests that hold the DCNv2 restatement (oracle/dcn_ref.py) without
torchvision: the op is third-party and absent, so these are its only pin."""
import torch
import torch.nn.functional as F

from oracle.dcn_ref import deform_conv2d


def _rand(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def test_zero_offset_unit_mask_is_conv2d():
    x, w, b = _rand(2, 8, 13, 17), _rand(6, 8, 3, 3, seed=1), _rand(6, seed=2)
    off = torch.zeros(2, 18, 13, 17)
    out = deform_conv2d(x, off, w, b, (1, 1), (1, 1), (1, 1), torch.ones(2, 9, 13, 17))
    torch.testing.assert_close(out, F.conv2d(x, w, b, 1, 1), rtol=1e-5, atol=1e-5)


def test_strided_dilated_zero_offset_is_conv2d():
    x, w = _rand(1, 4, 16, 20), _rand(5, 4, 3, 3, seed=1)
    ref = F.conv2d(x, w, None, stride=2, padding=2, dilation=2)
    off = torch.zeros(1, 18, *ref.shape[-2:])
    out = deform_conv2d(x, off, w, None, (2, 2), (2, 2), (2, 2), None)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)


def test_integer_offset_is_shifted_conv():
    x, w = _rand(1, 4, 12, 15), _rand(3, 4, 3, 3, seed=1)
    dy, dx = 2, -3
    off = torch.zeros(1, 18, 12, 15)
    off[:, 0::2] = dy
    off[:, 1::2] = dx
    out = deform_conv2d(x, off, w, None, (1, 1), (1, 1), (1, 1), None)
    # sampling x at (y+i-1+dy, x+j-1+dx), zero outside the image == plain conv over a
    # zero-padded canvas read at displaced positions
    P = 5
    full = F.conv2d(F.pad(x, (P, P, P, P)), w)
    exp = full[:, :, P - 1 + dy:P - 1 + dy + 12, P - 1 + dx:P - 1 + dx + 15]
    torch.testing.assert_close(out, exp, rtol=1e-5, atol=1e-5)


def test_offset_channel_order_dy_then_dx_per_tap():
    # only tap k=5 (i=1, j=2) is displaced, vertically by +1: moves that tap one row down
    x, w = _rand(1, 2, 9, 9), torch.zeros(1, 2, 3, 3)
    w[0, :, 1, 2] = 1.0
    off = torch.zeros(1, 18, 9, 9)
    off[:, 2 * 5] = 1.0
    out = deform_conv2d(x, off, w, None, (1, 1), (1, 1), (1, 1), None)
    # tap (1,2) samples (y+0+1, x+1)
    exp = torch.zeros(1, 1, 9, 9)
    exp[:, 0, :8, :8] = x[:, :, 1:, 1:].sum(1)
    torch.testing.assert_close(out, exp, rtol=1e-6, atol=1e-6)


def test_mask_is_linear_per_tap():
    x, w = _rand(1, 3, 8, 8), _rand(2, 3, 3, 3, seed=1)
    off = _rand(1, 18, 8, 8, seed=2)
    m1, m2 = torch.rand(1, 9, 8, 8), torch.rand(1, 9, 8, 8)
    f = lambda m: deform_conv2d(x, off, w, None, (1, 1), (1, 1), (1, 1), m)
    torch.testing.assert_close(f(m1 + 2 * m2), f(m1) + 2 * f(m2), rtol=1e-4, atol=1e-5)


def test_all_out_of_range_gives_bias():
    x, w, b = _rand(1, 3, 6, 6), _rand(4, 3, 3, 3, seed=1), _rand(4, seed=2)
    off = torch.full((1, 18, 6, 6), 100.0)
    out = deform_conv2d(x, off, w, b, (1, 1), (1, 1), (1, 1), torch.ones(1, 9, 6, 6))
    torch.testing.assert_close(out, b.view(1, 4, 1, 1).expand(1, 4, 6, 6))


def test_half_pixel_is_mean_of_integer_neighbours():
    x, w = _rand(1, 3, 10, 10), _rand(2, 3, 3, 3, seed=1)
    def run(dx):
        off = torch.zeros(1, 18, 10, 10)
        off[:, 1::2] = dx
        return deform_conv2d(x, off, w, None, (1, 1), (1, 1), (1, 1), None)
    torch.testing.assert_close(run(0.5), 0.5 * (run(0.0) + run(1.0)), rtol=1e-5, atol=1e-5)


def test_border_rule_minus_one_exclusive():
    # h in (-1, 0): only the h_high row contributes with weight lh; h == -1 exactly: zero
    x = torch.ones(1, 1, 4, 4)
    w = torch.zeros(1, 1, 3, 3)
    w[0, 0, 1, 1] = 1.0
    off = torch.zeros(1, 18, 4, 4)
    off[:, 8] = -0.25                       # centre tap dy
    out = deform_conv2d(x, off, w, None, (1, 1), (1, 1), (1, 1), None)
    assert torch.allclose(out[0, 0, 0], torch.full((4,), 0.75))
    assert torch.allclose(out[0, 0, 1:], torch.ones(3, 4))
    off[:, 8] = -1.0
    out = deform_conv2d(x, off, w, None, (1, 1), (1, 1), (1, 1), None)
    assert torch.all(out[0, 0, 0] == 0) and torch.all(out[0, 0, 1:] == 1)


def test_random_offsets_equal_an_independent_grid_sample_formulation():
    """Cross-check of the restated bilinear core against a DIFFERENT implementation of the same published semantics:
    torch's own `F.grid_sample(mode="bilinear", padding_mode="zeros", align_corners=True)` samples each tap's positions
    (a corner outside the image contributes zero; everything beyond (-1, H) x (-1, W) is zero - torchvision's rule), the
    mask and the weight contraction are applied with einsum.  Random fractional offsets of several pixels, positions on
    and beyond every border, float64: the two agree to rounding.  torchvision itself is absent, so this is
    the strongest pin available for `deform_conv2d`'s arithmetic."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(11)
    B, C, Co, H, W = 2, 8, 5, 13, 17
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
    off = torch.randn(B, 18, H, W, generator=g, dtype=torch.float64) * 3.0
    off[:, :, 0, :] -= 2.5                                   # rows / columns that sample above / left of / beyond the image
    off[:, :, :, -1] += 2.5
    off[0, :, 5, 5] = torch.tensor([-6.0, -6.0] * 9, dtype=torch.float64)       # far outside
    off[1, 0::2, 7, 3] = -0.5                                # exactly half a pixel
    mask = torch.rand(B, 9, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Co, C, 3, 3, generator=g, dtype=torch.float64)
    b = torch.randn(Co, generator=g, dtype=torch.float64)
    ref = deform_conv2d(x, off, w, b, (1, 1), (1, 1), (1, 1), mask)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    out = b.view(1, Co, 1, 1).expand(B, Co, H, W).clone()
    for k in range(9):
        i, j = divmod(k, 3)
        py = ys - 1 + i + off[:, 2 * k]                      # (B,H,W): channel 2k = dy, 2k+1 = dx of tap k = 3i + j
        px = xs - 1 + j + off[:, 2 * k + 1]
        grid = torch.stack([2 * px / (W - 1) - 1, 2 * py / (H - 1) - 1], dim=-1)
        s = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True)     # (B,C,H,W)
        out += torch.einsum("oc,bchw->bohw", w[:, :, i, j], s * mask[:, k:k + 1])
    assert float((out - ref).abs().max()) < 1e-11 * float(ref.abs().max() + 1)
