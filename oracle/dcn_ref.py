"""Oracle: modulated deformable convolution (DCNv2), CPU fp32.  TEST INFRASTRUCTURE.

Restates the documented semantics of ``torchvision.ops.deform_conv2d`` - the one
third-party op on the reference hot path (call site
/root/reference/src/lib/model/networks/dla.py:461-470).  torchvision is NOT in
/root/reference nor in this image and the reference does not pin its version
(requirements.txt:1-12), so this restatement is PARITY UNPINNED against the
real op; it is held by the known-answer tests in tests/test_oracle_dcn.py - and cross-checked there against an
independent formulation on torch's own F.grid_sample (float64, random offsets on and beyond every border) -
(zero offset == conv2d, integer offset == shifted conv, mask linearity,
all-out-of-range == bias, half-pixel == mean of neighbours).

Semantics restated (torchvision public docs + deform_conv2d kernel comments):
  * offset has 2*kh*kw channels per offset group; channel 2*k is the vertical
    (dy) and 2*k+1 the horizontal (dx) displacement of tap k = i*kw + j.
  * tap k of output pixel (y, x) samples input at
      h = y*stride_h - pad_h + i*dil_h + dy,   w = x*stride_w - pad_w + j*dil_w + dx
  * bilinear sample: 0 if h <= -1 or h >= H or w <= -1 or w >= W; otherwise the
    four neighbours (floor / floor+1), each contributing only when it lies
    inside the image; weights (1-lh)(1-lw), (1-lh)lw, lh(1-lw), lh*lw.
  * the sample is multiplied by mask[k] (already activated by the caller),
    contracted with weight (Cout, Cin, kh, kw), bias added.
"""
import torch


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def bilinear_columns(inp, offset, mask, kh, kw, stride, padding, dilation):
    """Return the sampled, mask-modulated columns (B, Cin, kh*kw, Ho, Wo)."""
    B, C, H, W = inp.shape
    sh, sw = stride
    ph, pw = padding
    dh, dw = dilation
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    assert offset.shape == (B, 2 * kh * kw, Ho, Wo), offset.shape
    dt = inp.dtype
    ys = torch.arange(Ho, dtype=dt).view(1, Ho, 1) * sh - ph
    xs = torch.arange(Wo, dtype=dt).view(1, 1, Wo) * sw - pw
    flat = inp.reshape(B, C, H * W)
    cols = []
    for i in range(kh):
        for j in range(kw):
            k = i * kw + j
            h = ys + i * dh + offset[:, 2 * k]
            w = xs + j * dw + offset[:, 2 * k + 1]
            inside = (h > -1) & (h < H) & (w > -1) & (w < W)
            h_low = torch.floor(h)
            w_low = torch.floor(w)
            lh = h - h_low
            lw = w - w_low
            hh = 1 - lh
            hw = 1 - lw
            h_low = h_low.long()
            w_low = w_low.long()
            h_high = h_low + 1
            w_high = w_low + 1

            def corner(hi, wi, ok):
                ok = ok & inside
                idx = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).view(B, 1, Ho * Wo)
                v = torch.gather(flat, 2, idx.expand(B, C, Ho * Wo)).view(B, C, Ho, Wo)
                return v * ok.view(B, 1, Ho, Wo).to(dt)

            v1 = corner(h_low, w_low, (h_low >= 0) & (w_low >= 0))
            v2 = corner(h_low, w_high, (h_low >= 0) & (w_high <= W - 1))
            v3 = corner(h_high, w_low, (h_high <= H - 1) & (w_low >= 0))
            v4 = corner(h_high, w_high, (h_high <= H - 1) & (w_high <= W - 1))
            w1 = (hh * hw).unsqueeze(1)
            w2 = (hh * lw).unsqueeze(1)
            w3 = (lh * hw).unsqueeze(1)
            w4 = (lh * lw).unsqueeze(1)
            val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
            if mask is not None:
                val = val * mask[:, k].unsqueeze(1)
            cols.append(val)
    return torch.stack(cols, dim=2), Ho, Wo


def deform_conv2d(input, offset, weight, bias=None, stride=(1, 1), padding=(0, 0),
                  dilation=(1, 1), mask=None):
    """Same signature as torchvision.ops.deform_conv2d (groups=1, offset groups=1)."""
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    Cout, Cin, kh, kw = weight.shape
    B = input.shape[0]
    assert input.shape[1] == Cin, "only groups=1 is on the reference path"
    cols, Ho, Wo = bilinear_columns(input, offset, mask, kh, kw, stride, padding, dilation)
    cols = cols.reshape(B, Cin * kh * kw, Ho * Wo)
    out = torch.matmul(weight.reshape(1, Cout, Cin * kh * kw), cols).view(B, Cout, Ho, Wo)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out
