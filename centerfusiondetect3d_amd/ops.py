"""Operator-level host wrappers over the C ABI (include/cf_hip.h).

Each function mirrors one operator the reference dispatches on the hot path (see the header for
the file:line each replaces).  Tensors are torch CUDA(=HIP) tensors used purely as device memory;
activations are NHWC fp32 (a (B,H,W,C) contiguous tensor).  Nothing here falls back to torch ops.
"""
import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import (ACT_NONE, ACT_RELU, ACT_SIGMOID_CLAMP, ACT_RAW_AND_SIGDEPTH, LAYOUT_NHWC,
                   LAYOUT_NCHW, LAYOUT_NHWC_SPLIT_BF16)
from .packing import PackedConv, PackedDcn, MX_ROW as packing_MX_ROW


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.CfHipError("libcfhip operators need device tensors (no CPU path)")


DEFAULT_IN_SCALE = 16.0     # activation pre-scale of the f16x3 / mx kernels (cf_f16x3.h: ASCALE); 65504 / 16 = 4094 is where they clamp
F16_MAX = 65504.0


def in_scale_for(absmax, headroom=8.0):
    """The activation pre-scale (a power of two) for a layer whose inputs reach `absmax`: the default 16 while that leaves a
    factor 4 to the fp16 limit, otherwise the largest power of two that keeps absmax * scale <= 65504 / headroom."""
    a = float(absmax)
    if not (a == a) or a == float("inf"):
        raise _lib.CfHipError("activation range is not finite (NaN / inf in a layer input)")
    if a * DEFAULT_IN_SCALE <= F16_MAX / 4.0:
        return DEFAULT_IN_SCALE
    import math
    return 2.0 ** math.floor(math.log2(F16_MAX / headroom / a))


def conv_args(pc: PackedConv, srcs: Sequence[torch.Tensor], src_strides: Sequence[int], B, H, W,
              out: torch.Tensor, out_stride: int, act=ACT_NONE, residual=None, res_stride=0,
              layout=LAYOUT_NHWC, out2=None, out_offset=0, precise=True, elem_bytes=4, in_scale=None) -> _lib.ConvArgs:
    """Build (and return for reuse) the argument block of one fused convolution.  in_scale (f16x3 kernels): the layer's
    activation pre-scale, a power of two (None = the default 16); out_scale follows it."""
    a = _lib.ConvArgs()
    for i, (s, c) in enumerate(zip(srcs, src_strides)):
        a.src[i] = s.data_ptr()
        a.src_c[i] = c
    a.n_src = len(srcs)
    a.B, a.H, a.W = B, H, W
    a.Ho = (H + 2 * pc.pad - pc.kh) // pc.stride + 1
    a.Wo = (W + 2 * pc.pad - pc.kh) // pc.stride + 1
    a.stride = pc.stride
    a.weight, a.slots, a.bias = pc.weight.data_ptr(), pc.slots.data_ptr(), pc.bias.data_ptr()
    a.K_pad, a.N, a.N_pad = pc.k_pad, pc.n, pc.n_pad
    a.residual = _lib.ptr(residual)
    a.res_stride = res_stride
    a.out = out.data_ptr() + elem_bytes * out_offset
    a.out2 = _lib.ptr(out2)
    a.out_stride, a.out_layout, a.act = out_stride, layout, act
    a.precise = int(bool(precise))
    a.out_scale = float(getattr(pc, "out_scale", 0.0))        # 2^-(s+4): the packer's figure at the default pre-scale
    if in_scale is not None and float(in_scale) != DEFAULT_IN_SCALE:
        a.in_scale = float(in_scale)
        a.out_scale = a.out_scale * DEFAULT_IN_SCALE / float(in_scale)
    return a


def absmax(x, out=None):
    """max |x| of an fp32 device tensor (contiguous, or a 2-D row-strided view) as a 1-element device tensor: NaN if any
    element is NaN, inf if any is infinite (cf_absmax_f32).  No host sync."""
    _need_cuda(x)
    if x.dtype != torch.float32:
        raise _lib.CfHipError("absmax: float32 only")
    if x.is_contiguous():
        M, Cc, stride = x.numel() // max(1, x.shape[-1] if x.dim() else 1), (x.shape[-1] if x.dim() else 1), (x.shape[-1] if x.dim() else 1)
        if Cc % 4 or x.data_ptr() % 16:
            M, Cc, stride = 1, x.numel(), x.numel()        # (one long row: the element-wise path)
    elif x.dim() == 2 and x.stride(1) == 1:
        M, Cc, stride = x.shape[0], x.shape[1], x.stride(0)
    else:
        raise _lib.CfHipError("absmax: contiguous tensor or a row-strided 2-D view")
    if out is None:
        out = torch.empty(1, device=x.device, dtype=torch.float32)
    if x.numel() == 0:
        out.zero_()
        return out
    _lib.check(_lib.load().cf_absmax_f32(x.data_ptr(), M, Cc, stride, out.data_ptr(), _lib.stream_ptr()), "cf_absmax_f32")
    return out


def run_conv_f16(a: _lib.ConvArgs, patch=False):
    if patch:
        _lib.check(_lib.load().cf_conv3x3_f16x3(C.byref(a), _lib.stream_ptr()), "cf_conv3x3_f16x3")
    else:
        _lib.check(_lib.load().cf_conv2d_f16x3(C.byref(a), _lib.stream_ptr()), "cf_conv2d_f16x3")


def conv2d_f16x3(pc: PackedConv, srcs, B, H, W, act=ACT_NONE, residual=None, out=None, patch=None):
    """fp32 NHWC in / out, split-fp16 products (packing.pack_conv_f16).  patch: use the LDS-patch
    3x3 kernel (default: whenever the packing allows it)."""
    _need_cuda(*srcs, residual)
    Ho = (H + 2 * pc.pad - pc.kh) // pc.stride + 1
    Wo = (W + 2 * pc.pad - pc.kh) // pc.stride + 1
    if out is None:
        out = torch.empty((B, Ho, Wo, pc.n), device=srcs[0].device, dtype=torch.float32)
    a = conv_args(pc, srcs, [s.shape[-1] for s in srcs], B, H, W, out, out.shape[-1], act, residual,
                  residual.shape[-1] if residual is not None else 0, LAYOUT_NHWC, None, 0, False)
    run_conv_f16(a, pc.patch if patch is None else patch)
    return out


def conv3x3_proj_f16x3(pc: PackedConv, t, pooled, act=ACT_RELU, out=None):
    """BasicBlock conv2 + the Tree's `project` of the pooled level input in ONE launch (cf_conv3x3_proj_f16x3; pc from
    packing.pack_conv_f16(proj=...)): out = act(conv3x3(t) + project(pooled) + biases)."""
    _need_cuda(t, pooled)
    assert pc.proj_k > 0, "weights packed without a projection"
    B, H, W, _ = t.shape
    if out is None:
        out = torch.empty((B, H, W, pc.n), device=t.device, dtype=torch.float32)
    a = conv_args(pc, [t, pooled], [t.shape[-1], pooled.shape[-1]], B, H, W, out, out.shape[-1], act, None, 0,
                  LAYOUT_NHWC, None, 0, False)
    ch = (C.c_int32 * 2)(*[int(c) for c in pc.real_cin])
    _lib.check(_lib.load().cf_conv3x3_proj_f16x3(C.byref(a), ch, _lib.stream_ptr()), "cf_conv3x3_proj_f16x3")
    return out


def conv3x3_root_f16x3(pc2: PackedConv, pc_root: PackedConv, t, x1, children=(), act_root=ACT_RELU, x2_out=None):
    """One-level Tree tail: x2 = ReLU(conv3x3(t) + x1), out = act(Root([x2, x1, *children])) as ONE launch where the shape
    allows (cf_conv3x3_root_f16x3), the two launches otherwise - same bits.  -> (out, x2 buffer: written only on the fallback)."""
    _need_cuda(t, x1, *children)
    B, H, W, _ = t.shape
    x2 = torch.empty((B, H, W, pc2.n), device=t.device, dtype=torch.float32) if x2_out is None else x2_out
    out = torch.empty((B, H, W, pc_root.n), device=t.device, dtype=torch.float32)
    a = conv_args(pc2, [t], [t.shape[-1]], B, H, W, x2, x2.shape[-1], ACT_RELU, x1, x1.shape[-1], LAYOUT_NHWC, None, 0, False)
    srcs = [x2, x1, *children]
    r = conv_args(pc_root, srcs, [s.shape[-1] for s in srcs], B, H, W, out, out.shape[-1], act_root, None, 0,
                  LAYOUT_NHWC, None, 0, False)
    ch = (C.c_int32 * len(srcs))(*[int(c) for c in pc_root.real_cin])
    _lib.check(_lib.load().cf_conv3x3_root_f16x3(C.byref(a), C.byref(r), ch, _lib.stream_ptr()), "cf_conv3x3_root_f16x3")
    return out, x2


def run_conv(a: _lib.ConvArgs):
    _lib.check(_lib.load().cf_conv2d_fused(C.byref(a), _lib.stream_ptr()), "cf_conv2d_fused")


def conv2d_fused(pc: PackedConv, srcs, B, H, W, act=ACT_NONE, residual=None, layout=LAYOUT_NHWC,
                 out=None, out2=None, precise=True):
    """Convenience form: allocates the output.  srcs: NHWC tensors (B,H,W,Ci)."""
    _need_cuda(*srcs, residual)
    Ho = (H + 2 * pc.pad - pc.kh) // pc.stride + 1
    Wo = (W + 2 * pc.pad - pc.kh) // pc.stride + 1
    dev = srcs[0].device
    if out is None:
        shape = (B, Ho, Wo, pc.n) if layout == LAYOUT_NHWC else (B, pc.n, Ho, Wo)
        out = torch.empty(shape, device=dev, dtype=torch.float32)
    if act == ACT_RAW_AND_SIGDEPTH and out2 is None:
        out2 = torch.empty_like(out)
    a = conv_args(pc, srcs, [s.shape[-1] for s in srcs], B, H, W, out, pc.n, act, residual,
                  residual.shape[-1] if residual is not None else 0, layout, out2, 0, precise)
    run_conv(a)
    return (out, out2) if act == ACT_RAW_AND_SIGDEPTH else out


def run_conv_bf16(a: _lib.ConvArgs):
    _lib.check(_lib.load().cf_conv2d_bf16x3(C.byref(a), _lib.stream_ptr()), "cf_conv2d_bf16x3")


def split_bf16(x, channels=None, cs=None, out=None):
    """fp32 NHWC (B,H,W,S) -> split-bf16 (B,H,W,2,Cs) stored as a bf16 tensor."""
    _need_cuda(x)
    B, H, W, S = x.shape
    Cc = channels or S
    Cs = cs or ((Cc + 7) // 8) * 8
    if out is None:
        out = torch.empty((B, H, W, 2, Cs), device=x.device, dtype=torch.bfloat16)
    _lib.check(_lib.load().cf_split_bf16(x.data_ptr(), out.data_ptr(), B * H * W, Cc, S, Cs,
                                         _lib.stream_ptr()), "cf_split_bf16")
    return out


def conv2d_bf16x3(pc: PackedConv, srcs, B, H, W, act=ACT_NONE, layout=LAYOUT_NHWC_SPLIT_BF16,
                  out=None, out2=None):
    """srcs: split-bf16 tensors (B,H,W,2,Cs).  Returns split-bf16 (B,Ho,Wo,2,N) or fp32 NCHW."""
    _need_cuda(*srcs)
    Ho = (H + 2 * pc.pad - pc.kh) // pc.stride + 1
    Wo = (W + 2 * pc.pad - pc.kh) // pc.stride + 1
    dev = srcs[0].device
    if out is None:
        if layout == LAYOUT_NHWC_SPLIT_BF16:
            out = torch.empty((B, Ho, Wo, 2, pc.n), device=dev, dtype=torch.bfloat16)
        else:
            out = torch.empty((B, pc.n, Ho, Wo), device=dev, dtype=torch.float32)
    if act == ACT_RAW_AND_SIGDEPTH and out2 is None:
        out2 = torch.empty_like(out)
    a = conv_args(pc, srcs, [s.shape[-1] for s in srcs], B, H, W, out, pc.n, act, None, 0, layout,
                  out2, 0, False)
    run_conv_bf16(a)
    return (out, out2) if act == ACT_RAW_AND_SIGDEPTH else out


def head_tail_args(x, x_stride, B, H, W, heads):
    """heads: list of dicts {c_base, w_hidden:[frag tensors], b_hidden:[f32 tensors], w_out, b_out,
    n_out, act, out (NCHW tensor or None), out2}.  Outputs may be patched later (a.out[i] = ptr)."""
    a = _lib.HeadTailArgs()
    a.x, a.x_stride, a.B, a.H, a.W = x.data_ptr(), x_stride, B, H, W
    a.n_heads = len(heads)
    a.n_hidden = len(heads[0]["w_hidden"])
    for i, hd in enumerate(heads):
        assert len(hd["w_hidden"]) == a.n_hidden
        for l, (w, b) in enumerate(zip(hd["w_hidden"], hd["b_hidden"])):
            a.w_hidden[i][l], a.b_hidden[i][l] = w.data_ptr(), b.data_ptr()
        a.w_out[i], a.b_out[i] = hd["w_out"].data_ptr(), hd["b_out"].data_ptr()
        a.out[i] = _lib.ptr(hd.get("out"))
        a.out2[i] = _lib.ptr(hd.get("out2"))
        a.c_base[i], a.n_out[i], a.act[i] = hd["c_base"], hd["n_out"], hd["act"]
    return a


def head_fused_args(srcs, src_strides, slots, k_pad, B, H, W, heads, layout3x3=None):
    """heads: as head_tail_args plus w_first (fragment-packed [8][K_pad/16]...) and b_first (256 f32).
    layout3x3 (default: every head carries w_out_perm): the slots are pack_conv_bf16's canonical order for
    [feat 64 (, pc_hm 8)] sources, so the 2-D patch kernel may take the launch."""
    f = _lib.HeadFusedArgs()
    t = head_tail_args(srcs[0], 256, B, H, W, [dict(hd, c_base=0) for hd in heads])
    C.memmove(C.byref(f.tail), C.byref(t), C.sizeof(t))
    for i, (s, c) in enumerate(zip(srcs, src_strides)):
        f.src[i], f.src_c[i] = s.data_ptr(), c
    f.n_src = len(srcs)
    f.slots, f.K_pad = (slots.data_ptr() if slots is not None else None), int(k_pad or 0)   # (mx: no slot table)
    for i, hd in enumerate(heads):
        f.w_first[i], f.b_first[i] = hd["w_first"].data_ptr(), hd["b_first"].data_ptr()
        if hd.get("w_out_perm") is not None:
            f.w_out_perm[i] = hd["w_out_perm"].data_ptr()
    f.layout3x3 = int(all(hd.get("w_out_perm") is not None for hd in heads)) if layout3x3 is None else int(layout3x3)
    f.mfma16 = int(all(bool(hd.get("mfma16")) for hd in heads))     # fragments packed for the 16x16x32 shape (the C
    # side refuses them on a launch that does not take the 3x3 patch kernel: nothing else can read them)
    f.mx = int(all(hd.get("first_scale") is not None for hd in heads))   # packing.pack_head_first_mx streams: srcs[0] = mx rows
    if f.mx:
        for i, hd in enumerate(heads):
            f.first_scale[i] = float(hd["first_scale"])
    return f


def pack_feat_mx(feat, out=None, scale=None):
    """feat (..., C >= 64) fp32 NHWC (the first 64 channels are the feature map) -> (..., 272) uint8 rows for
    cf_head_fused with mx = 1 (include/cf_hip.h: cf_pack_feat_mx; scale: the rows' power-of-two pre-scale, None = 16)."""
    _need_cuda(feat)
    assert feat.dtype == torch.float32 and feat.is_contiguous() and feat.shape[-1] >= 64
    M = feat.numel() // feat.shape[-1]
    if out is None:
        out = torch.empty(feat.shape[:-1] + (packing_MX_ROW,), device=feat.device, dtype=torch.uint8)
    _lib.check(_lib.load().cf_pack_feat_mx_scaled(feat.data_ptr(), feat.shape[-1], out.data_ptr(), M,
                                                  float(scale or DEFAULT_IN_SCALE), _lib.stream_ptr()), "cf_pack_feat_mx")
    return out


def run_head_fused(f):
    _lib.check(_lib.load().cf_head_fused(C.byref(f), _lib.stream_ptr()), "cf_head_fused")


def run_head_tail(a):
    _lib.check(_lib.load().cf_head_tail(C.byref(a), _lib.stream_ptr()), "cf_head_tail")


def dcn_args(pd: PackedDcn, x, offmask, om_stride, B, H, W, out, out_stride, act=ACT_RELU,
             precise=True, out_split=None, workspace=None, out_mx=None, in_scale=None, mx_scale=None):
    """in_scale: the activation pre-scale of `x` (f16x3 kernel; None = 16); mx_scale: the pre-scale of the out_mx rows."""
    a = _lib.DcnArgs()
    a.x, a.offmask, a.om_stride = x.data_ptr(), offmask.data_ptr(), om_stride
    a.B, a.H, a.W, a.C = B, H, W, pd.c
    a.weight, a.bias, a.N, a.N_pad = pd.weight.data_ptr(), pd.bias.data_ptr(), pd.n, pd.n_pad
    a.out, a.out_stride, a.act = out.data_ptr(), out_stride, act
    a.precise = int(bool(precise))
    a.out_scale = float(getattr(pd, "out_scale", 0.0))
    if in_scale is not None and float(in_scale) != DEFAULT_IN_SCALE:
        a.in_scale = float(in_scale)
        a.out_scale = a.out_scale * DEFAULT_IN_SCALE / float(in_scale)
    if mx_scale is not None and float(mx_scale) != DEFAULT_IN_SCALE:
        a.mx_scale = float(mx_scale)
    if out_split is not None:            # (B,H,W,2,Cs) bf16: split copy for the head kernels (f16x3 kernel only)
        a.out_split_bf16, a.split_stride = out_split.data_ptr(), out_split.shape[-1]
    if out_mx is not None:               # (B,H,W,272) uint8: the mx rows of the heads' first layer (f16x3 kernel, N = 64)
        a.out_mx = out_mx.data_ptr()
    if workspace is not None:            # K split on small maps (cf_dcn_v2_workspace_bytes)
        a.workspace = workspace.data_ptr()
        a.workspace_bytes = workspace.numel() * workspace.element_size()
    return a


def run_dcn(a: _lib.DcnArgs):
    fn = _lib.load().cf_dcn_v2_f16x3 if a.out_scale > 0 else _lib.load().cf_dcn_v2_fused
    _lib.check(fn(C.byref(a), _lib.stream_ptr()), "cf_dcn_v2")


def dcn_v2_fused(pd: PackedDcn, x, offmask, act=ACT_RELU, precise=True, k_split=True):
    """x (B,H,W,C) NHWC, offmask (B,H,W,S>=27) NHWC raw conv_offset_mask output."""
    _need_cuda(x, offmask)
    B, H, W, _ = x.shape
    out = torch.empty((B, H, W, pd.n), device=x.device, dtype=torch.float32)
    ws = None
    if k_split and getattr(pd, "out_scale", 0.0) > 0:
        nbytes = _lib.load().cf_dcn_v2_workspace_bytes(B, H, W, pd.c, pd.n_pad)
        ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8) if nbytes else None
    a = dcn_args(pd, x, offmask, offmask.shape[-1], B, H, W, out, pd.n, act, precise, workspace=ws)
    run_dcn(a)
    return out


def nchw_to_nhwc(x, out=None, out_offset=0):
    """(B,C,H,W) fp32 -> (B,H,W,S) NHWC, written at channel `out_offset` of `out` (default: a fresh (B,H,W,C))."""
    _need_cuda(x, out)
    B, Cc, H, W = x.shape
    if out is None:
        out = torch.empty((B, H, W, Cc), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().cf_nchw_to_nhwc(x.data_ptr(), out.data_ptr(), B, Cc, H, W, out.shape[-1], out_offset,
                                           _lib.stream_ptr()), "cf_nchw_to_nhwc")
    return out


_DCN_PACKS = {}       # id(weight) -> (weakref to weight, weight stamp, weakref to bias or None, bias stamp, PackedDcn)
_DCN_PACK_KEYS = 64   # LRU bound
_DCN_PACK_LOCK = __import__("threading").Lock()   # (the operator may be called from several host threads)
_DCN_PACK_VERIFY = True


def set_dcn_pack_verify(on=True):
    """`deform_conv2d` keeps the packed form of each weight it has seen.  An in-place write through `weight.data`
    (`w.data.copy_()`, an EMA swap, a hand-written checkpoint load) bumps no version counter autograd can see, so with
    verification ON (the default) every call also compares a 64-bit checksum of the live weight and bias bits with the one
    taken when the entry was packed - one small reduction and one device->host scalar per call.  Turn it off for a serving
    loop whose weights are frozen (and call `clear_dcn_pack_cache()` after any manual write).  -> previous setting."""
    global _DCN_PACK_VERIFY
    prev, _DCN_PACK_VERIFY = _DCN_PACK_VERIFY, bool(on)
    return prev


_DCN_RANGE_CHECK = True


def set_dcn_range_check(on=True):
    """`deform_conv2d` evaluates its products from fp16-split operands after a power-of-two pre-scale of the input (16 by
    default: |input| up to 4094).  With the check ON (the default) every call measures max |input| (cf_absmax_f32, one
    device->host scalar) and picks the pre-scale for it, so any finite fp32 input is accepted as torchvision's operator
    accepts it; a NaN / inf input raises.  OFF: the default pre-scale, no sync - inputs beyond 4094 are then CLAMPED silently;
    only for callers that know their range.  Inside a stream capture the check cannot run (no sync): default pre-scale.
    -> previous setting."""
    global _DCN_RANGE_CHECK
    prev, _DCN_RANGE_CHECK = _DCN_RANGE_CHECK, bool(on)
    return prev


def clear_dcn_pack_cache():
    """Drop every packed weight `deform_conv2d` holds (they are rebuilt on the next call)."""
    with _DCN_PACK_LOCK:
        _DCN_PACKS.clear()


def _stamp(t):
    """what autograd and the allocator can tell about a tensor's contents without reading them"""
    return (t._version, t.data_ptr(), tuple(t.shape), t.dtype)


def _checksum(weight, bias):
    """64-bit sum of the raw bits (order-independent, exact: integer arithmetic) - a device scalar."""
    def bits(t):
        t = t.detach().contiguous()
        return t.view(torch.int32 if t.element_size() == 4 else torch.int16 if t.element_size() == 2 else torch.int64)
    c = bits(weight).sum(dtype=torch.int64)
    if bias is not None:
        c = c * 1000003 + bits(bias).sum(dtype=torch.int64)
    return c


def _packed_dcn(weight, bias):
    """Pack (split fp16 hi / lo, MFMA fragment order: packing.pack_dcn_f16) ONCE per weight tensor OBJECT and content -
    the reference calls the operator with the same nn.Parameter every forward (dla.py:464-465).  The entry holds weak
    references and is valid only while they still point at the very tensors passed in (a data pointer or an id() alone
    can be reused by another tensor after the first one is freed), their version counter, data pointer, shape and dtype
    are unchanged, and - `set_dcn_pack_verify` - their bits still add up to the checksum taken at packing time."""
    import weakref
    from . import packing
    with _DCN_PACK_LOCK:
        return _packed_dcn_locked(weight, bias, weakref, packing)


def _packed_dcn_locked(weight, bias, weakref, packing):
    key = id(weight)
    e = _DCN_PACKS.pop(key, None)
    verify = _DCN_PACK_VERIFY and weight.is_cuda and not torch.cuda.is_current_stream_capturing()
    csum = None
    if e is not None:
        wref, wst, bref, bst, pd, csum0 = e
        same_bias = (bref is None) if bias is None else (bref is not None and bref() is bias and bst == _stamp(bias))
        if not (wref() is weight and wst == _stamp(weight) and same_bias and pd.weight.device == weight.device):
            e = None
        elif verify:
            csum = int(_checksum(weight, bias).item())
            if csum0 is None:
                e = (wref, wst, bref, bst, pd, csum)      # packed during a capture: the checksum is taken now
            elif csum != csum0:
                e = None
    if e is None:
        w = weight.detach().float().cpu()
        b = torch.zeros(w.shape[0]) if bias is None else bias.detach().float().cpu()
        pd = packing.pack_dcn_f16(w, b).to(weight.device)
        if verify and csum is None:
            csum = int(_checksum(weight, bias).item())
        e = (weakref.ref(weight), _stamp(weight), None if bias is None else weakref.ref(bias),
             None if bias is None else _stamp(bias), pd, csum if verify else None)
    for k in [k for k, v in _DCN_PACKS.items() if v[0]() is None]:
        del _DCN_PACKS[k]                         # weights that no longer exist: free their packed copies now
    _DCN_PACKS[key] = e                           # re-inserted last: dict order is the LRU order
    while len(_DCN_PACKS) > _DCN_PACK_KEYS:
        _DCN_PACKS.pop(next(iter(_DCN_PACKS)))
    return e[4]


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else tuple(int(e) for e in v)


def deform_conv2d(input, offset, weight, bias=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1), mask=None):
    """Operator-level drop-in for `torchvision.ops.deform_conv2d` as the reference calls it
    (model/networks/dla.py:461-470) - same signature, same semantics: `input` (B,Cin,H,W),
    `offset` (B,18,H,W) with channel 2k = dy and 2k+1 = dx of tap k = 3i+j, `mask` (B,9,H,W) ALREADY activated (the
    caller's sigmoid; None = unmodulated DCNv1), raw `weight` (Cout,Cin,3,3) and `bias` (Cout) -> (B,Cout,H,W), all
    NCHW fp32 on the device.  `torchvision.ops.deform_conv2d = centerfusiondetect3d_amd.ops.deform_conv2d` lets the
    reference's own `DeformConv` module run on the HIP kernel unchanged.

    What runs: three layout launches (cf_nchw_to_nhwc: input, offset, mask -> the NHWC buffers the kernel reads),
    cf_dcn_v2_f16x3 with `mask_activated`, cf_nhwc_to_nchw.  The module path (DLASeg) never takes this route: there
    the maps stay NHWC and the mask logits go in raw.  Only the configuration the reference uses is implemented -
    3x3, stride 1, padding 1, dilation 1, one group, one offset group, Cin a multiple of 32; anything else raises
    NotImplementedError (no fallback)."""
    _need_cuda(input, offset, weight, bias, mask)
    if input.dim() != 4 or weight.dim() != 4:
        raise ValueError("deform_conv2d: input must be (B,Cin,H,W) and weight (Cout,Cin,kh,kw)")
    B, Cin, H, W = input.shape
    Cout, Cw, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or _pair(stride) != (1, 1) or _pair(padding) != (1, 1) or _pair(dilation) != (1, 1):
        raise NotImplementedError("deform_conv2d on the HIP path: 3x3, stride 1, padding 1, dilation 1 only "
                                  "(the one configuration of dla.py:385-472)")
    if Cw != Cin:
        raise NotImplementedError("deform_conv2d on the HIP path: groups == 1 only")
    if Cin % 32:
        raise NotImplementedError(f"deform_conv2d on the HIP path: Cin={Cin} must be a multiple of 32")
    if tuple(offset.shape) != (B, 18, H, W):
        if offset.dim() == 4 and offset.shape[1] % 18 == 0 and offset.shape[1] > 18:
            raise NotImplementedError("deform_conv2d on the HIP path: one offset group only")
        raise ValueError(f"deform_conv2d: offset must be {(B, 18, H, W)}, got {tuple(offset.shape)}")
    if mask is not None and tuple(mask.shape) != (B, 9, H, W):
        raise ValueError(f"deform_conv2d: mask must be {(B, 9, H, W)}, got {tuple(mask.shape)}")
    if bias is not None and tuple(bias.shape) != (Cout,):
        raise ValueError(f"deform_conv2d: bias must be ({Cout},), got {tuple(bias.shape)}")
    dev = input.device
    pd = _packed_dcn(weight, bias)
    x = nchw_to_nhwc(input.float().contiguous())
    om = torch.empty((B, H, W, 32), device=dev, dtype=torch.float32)
    nchw_to_nhwc(offset.float().contiguous(), om, 0)
    if mask is None:
        om[..., 18:27] = 1.0
    else:
        nchw_to_nhwc(mask.float().contiguous(), om, 18)
    out = torch.empty((B, H, W, pd.n_pad), device=dev, dtype=torch.float32)     # (any Cout: the row stride is N_pad)
    nbytes = _lib.load().cf_dcn_v2_workspace_bytes(B, H, W, pd.c, pd.n_pad)
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8) if nbytes else None
    in_scale = None
    if _DCN_RANGE_CHECK and not torch.cuda.is_current_stream_capturing():
        in_scale = in_scale_for(float(absmax(x).item()))          # (raises on NaN / inf)
    a = dcn_args(pd, x, om, 32, B, H, W, out, pd.n_pad, ACT_NONE, precise=True, workspace=ws, in_scale=in_scale)
    a.mask_activated = 1
    run_dcn(a)
    return nhwc_to_nchw(out, channels=pd.n)


def upsample_dw(x, weight_kkc, f, skip=None, out=None):
    _need_cuda(x, weight_kkc, skip)
    B, H, W, Cc = x.shape
    if out is None:
        out = torch.empty((B, H * f, W * f, Cc), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().cf_upsample_dw(x.data_ptr(), weight_kkc.data_ptr(), _lib.ptr(skip),
                                          out.data_ptr(), B, H, W, Cc, f, _lib.stream_ptr()),
               "cf_upsample_dw")
    return out


def maxpool2x2(x, out=None):
    _need_cuda(x)
    B, H, W, Cc = x.shape
    if out is None:
        out = torch.empty((B, H // 2, W // 2, Cc), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().cf_maxpool2x2(x.data_ptr(), out.data_ptr(), B, H, W, Cc,
                                         _lib.stream_ptr()), "cf_maxpool2x2")
    return out


def nchw_to_nhwc4(x, out=None):
    _need_cuda(x)
    B, Cc, H, W = x.shape
    if out is None:
        out = torch.empty((B, H, W, 4), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().cf_nchw_to_nhwc4(x.data_ptr(), out.data_ptr(), B, Cc, H, W,
                                            _lib.stream_ptr()), "cf_nchw_to_nhwc4")
    return out


def nhwc_to_nchw(x, channels=None, out=None):
    _need_cuda(x)
    B, H, W, S = x.shape
    Cc = channels or S
    if out is None:
        out = torch.empty((B, Cc, H, W), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().cf_nhwc_to_nchw(x.data_ptr(), out.data_ptr(), B, H, W, Cc, S,
                                           _lib.stream_ptr()), "cf_nhwc_to_nchw")
    return out


CHECKSUM_PARTS = 256     # CF_CHECKSUM_PARTS of include/cf_hip.h


def checksum64(t, out=None):
    """Position-weighted 64-bit checksum of a contiguous device tensor's bits (cf_checksum64) as CHECKSUM_PARTS partial sums
    over a fixed partition of its words -> (256,) int64 device tensor: equal bits <=> equal parts.  No host sync."""
    _need_cuda(t)
    nbytes = t.numel() * t.element_size()
    if not t.is_contiguous() or nbytes % 4:
        raise _lib.CfHipError("checksum64: contiguous tensor of a multiple of 4 bytes")
    if out is None:
        out = torch.empty(CHECKSUM_PARTS, device=t.device, dtype=torch.int64)
    assert out.numel() == CHECKSUM_PARTS and out.dtype == torch.int64 and out.is_contiguous()
    _lib.check(_lib.load().cf_checksum64(t.data_ptr(), nbytes // 4, out.data_ptr(), _lib.stream_ptr()), "cf_checksum64")
    return out


def topk_peaks(heat, K=100, nms=False, out=None, only_if_changed=None):
    """(B,C,H,W) NCHW scores -> scores (B,K) f32, inds (B,K) i32, classes (B,K) i32.
    nms: False / True (3x3 equality NMS first; True = the two-pass form, 1 = suppress on the fly).
    out: (scores, inds, classes) to write into; only_if_changed: a (2 * CHECKSUM_PARTS,) int64 device tensor [expected parts |
    actual parts] - the launch does nothing when they are equal (cf_topk_peaks_if_changed: `out` then keeps what it held) and
    computes the NMS'd peaks otherwise (one workgroup per image: the rare path)."""
    _need_cuda(heat)
    if not heat.is_contiguous():
        heat = heat.contiguous()
    B, Cc, H, W = heat.shape
    dev = heat.device
    if out is not None:
        scores, inds, classes = out
    else:
        scores = torch.empty((B, K), device=dev, dtype=torch.float32)
        inds = torch.empty((B, K), device=dev, dtype=torch.int32)
        classes = torch.empty((B, K), device=dev, dtype=torch.int32)
    lib = _lib.load()
    mode = int(nms) if nms in (0, 1, 2) and not isinstance(nms, bool) else (2 if nms else 0)
    size = lib.cf_topk_workspace_bytes_nms(B, Cc, H, W, K) if mode == 2 else lib.cf_topk_workspace_bytes(B, K)
    ws = torch.empty(max(1, size), device=dev, dtype=torch.uint8)
    if only_if_changed is not None:
        _lib.check(lib.cf_topk_peaks_if_changed(heat.data_ptr(), B, Cc, H, W, K, mode, scores.data_ptr(), inds.data_ptr(),
                                                classes.data_ptr(), ws.data_ptr(), only_if_changed.data_ptr(),
                                                _lib.stream_ptr()), "cf_topk_peaks_if_changed")
        return scores, inds, classes
    _lib.check(lib.cf_topk_peaks(heat.data_ptr(), B, Cc, H, W, K, mode,
                                 scores.data_ptr(), inds.data_ptr(), classes.data_ptr(),
                                 ws.data_ptr(), _lib.stream_ptr()), "cf_topk_peaks")
    return scores, inds, classes


def frustum_assoc(inds, depth, wh, dim, rot, calib, pc_dep, max_pc_dist=60.0, want_nhwc4=False,
                  pc_hm=None, pc_hm_nhwc4=None, pc_hm_split8=None):
    _need_cuda(inds, depth, wh, dim, rot, calib, pc_dep)
    B, _, H, W = pc_dep.shape
    K = inds.shape[1]
    dev = pc_dep.device
    if pc_hm is None:
        pc_hm = torch.empty((B, 3, H, W), device=dev, dtype=torch.float32)
    if want_nhwc4 and pc_hm_nhwc4 is None:
        pc_hm_nhwc4 = torch.empty((B, H, W, 4), device=dev, dtype=torch.float32)
    ts = [t if t.is_contiguous() else t.contiguous() for t in (depth, wh, dim, rot, calib, pc_dep)]
    _lib.check(_lib.load().cf_frustum_assoc(inds.data_ptr(), K, *(t.data_ptr() for t in ts), B, H,
                                            W, float(max_pc_dist), pc_hm.data_ptr(),
                                            _lib.ptr(pc_hm_nhwc4), _lib.ptr(pc_hm_split8),
                                            _lib.stream_ptr()),
               "cf_frustum_assoc")
    return (pc_hm, pc_hm_nhwc4) if want_nhwc4 else pc_hm


def topk_frustum(heat, depth, wh, dim, rot, calib, pc_dep, K=100, max_pc_dist=60.0, want_nhwc4=False, want_split8=False,
                 want_peaks=False):
    """cf_topk_frustum: top-K of the raw heat map + frustum association in two launches (the slice lists are merged in the
    association kernel's prologue) - the same pc_hm as topk_peaks(heat, K) followed by frustum_assoc, bit for bit.
    -> pc_hm (B,3,H,W) [, nhwc4 (B,H,W,4)] [, split8 (B,H,W,2,8) bf16] [, (scores, inds, classes)]"""
    _need_cuda(heat, depth, wh, dim, rot, calib, pc_dep)
    B, Cc, H, W = heat.shape
    dev = heat.device
    ts = [t if t.is_contiguous() else t.contiguous() for t in (heat, depth, wh, dim, rot, calib, pc_dep)]
    pc_hm = torch.empty((B, 3, H, W), device=dev, dtype=torch.float32)
    hm4 = torch.empty((B, H, W, 4), device=dev, dtype=torch.float32) if want_nhwc4 else None
    hm8 = torch.empty((B, H, W, 2, 8), device=dev, dtype=torch.bfloat16) if want_split8 else None
    peaks = (torch.empty((B, K), device=dev, dtype=torch.float32), torch.empty((B, K), device=dev, dtype=torch.int32),
             torch.empty((B, K), device=dev, dtype=torch.int32)) if want_peaks else (None, None, None)
    lib = _lib.load()
    ws = torch.empty(max(1, lib.cf_topk_workspace_bytes(B, K)), device=dev, dtype=torch.uint8)
    _lib.check(lib.cf_topk_frustum(ts[0].data_ptr(), Cc, K, *(t.data_ptr() for t in ts[1:]), B, H, W, float(max_pc_dist),
                                   pc_hm.data_ptr(), _lib.ptr(hm4), _lib.ptr(hm8), *(_lib.ptr(t) for t in peaks),
                                   ws.data_ptr(), _lib.stream_ptr()), "cf_topk_frustum")
    out = [pc_hm] + ([hm4] if want_nhwc4 else []) + ([hm8] if want_split8 else []) + ([peaks] if want_peaks else [])
    return out[0] if len(out) == 1 else tuple(out)


def _decode_args(scores, inds, classes, maps: dict, H, W, out_hw, norm2d, det):
    _need_cuda(scores, inds, classes)
    B, K = scores.shape
    a = _lib.DecodeArgs()
    a.scores, a.inds, a.classes = scores.data_ptr(), inds.data_ptr(), classes.data_ptr()
    keep = []
    for name in ("reg", "wh", "depth", "rot", "dim", "amodal", "att", "vel"):
        t = maps.get(name)
        if t is not None and not t.is_contiguous():
            t = t.contiguous()
        keep.append(t)
        setattr(a, name, _lib.ptr(t))
    a.B, a.K, a.H, a.W = B, K, H, W
    a.out_h, a.out_w, a.norm2d = int(out_hw[0]), int(out_hw[1]), int(bool(norm2d))
    a.det = _lib.ptr(det)
    return a, keep


def decode_gather(scores, inds, classes, maps: dict, H, W, out_hw, norm2d=False):
    """maps: optional NCHW tensors under reg/wh/depth/rot/dim/amodal/att/vel -> det (B,K,33)."""
    B, K = scores.shape
    det = torch.empty((B, K, 33), device=scores.device, dtype=torch.float32)
    a, _keep = _decode_args(scores, inds, classes, maps, H, W, out_hw, norm2d, det)
    _lib.check(_lib.load().cf_decode_gather(C.byref(a), _lib.stream_ptr()), "cf_decode_gather")
    return det


def decode_post(scores, inds, classes, maps: dict, H, W, out_hw, calib, trans_inv, norm2d=False, want_det=False):
    """cf_decode_post: decode rows and postProcess rows in one launch -> post (B,K,54) [, det (B,K,33)].
    calib (B,3,4) f32, trans_inv (2,3) f32 device (output map -> source image affine)."""
    _need_cuda(calib, trans_inv)
    B, K = scores.shape
    dev = scores.device
    det = torch.empty((B, K, 33), device=dev, dtype=torch.float32) if want_det else None
    post = torch.empty((B, K, 54), device=dev, dtype=torch.float32)
    a, _keep = _decode_args(scores, inds, classes, maps, H, W, out_hw, norm2d, det)
    if calib.dtype != torch.float32 or not calib.is_contiguous() or calib.numel() != B * 12:
        raise _lib.CfHipError("cf_decode_post: calib must be contiguous float32 (B,3,4)")
    if trans_inv.dtype != torch.float32 or not trans_inv.is_contiguous() or trans_inv.numel() != 6:
        raise _lib.CfHipError("cf_decode_post: trans_inv must be contiguous float32 (2,3)")
    _lib.check(_lib.load().cf_decode_post(C.byref(a), calib.data_ptr(), trans_inv.data_ptr(), post.data_ptr(),
                                          _lib.stream_ptr()), "cf_decode_post")
    return (post, det) if want_det else post


def serialize_nuscenes(post, trans_matrix, velocity_matrix, cs_rot=None, pose_rot=None, sample_ptr=None,
                       sample_frames=None, max_per_sample=500):
    """cf_serialize_nuscenes -> rows (B*K,12) f32, rotation (B*K,4) f64, order (S,max) i32, counts (S) i32
    (order / counts are None without the sample tables)."""
    _need_cuda(post, trans_matrix, velocity_matrix)
    B, K, w = post.shape
    dev = post.device
    for t, dt, n in ((post, torch.float32, B * K * 54), (trans_matrix, torch.float32, B * 16),
                     (velocity_matrix, torch.float32, B * 16)):
        if t.dtype != dt or not t.is_contiguous() or t.numel() != n:
            raise _lib.CfHipError("cf_serialize_nuscenes: wrong dtype / shape / non-contiguous input")
    a = _lib.SerializeArgs()
    a.post, a.B, a.K = post.data_ptr(), B, K
    a.trans_matrix, a.velocity_matrix = trans_matrix.data_ptr(), velocity_matrix.data_ptr()
    if (cs_rot is None) != (pose_rot is None):
        raise _lib.CfHipError("cf_serialize_nuscenes: cs_rot and pose_rot go together")
    for q in (cs_rot, pose_rot):
        if q is not None and (q.dtype != torch.float64 or not q.is_contiguous() or q.numel() != B * 4 or not q.is_cuda):
            raise _lib.CfHipError("cf_serialize_nuscenes: quaternions must be contiguous float64 (B,4) on the device")
    a.cs_rot, a.pose_rot = _lib.ptr(cs_rot), _lib.ptr(pose_rot)
    rows = torch.empty((B * K, 12), device=dev, dtype=torch.float32)
    rotation = torch.empty((B * K, 4), device=dev, dtype=torch.float64)
    a.rows, a.rotation = rows.data_ptr(), rotation.data_ptr()
    order = counts = None
    if sample_ptr is not None:
        _need_cuda(sample_ptr, sample_frames)
        if sample_ptr.dtype != torch.int32 or sample_frames.dtype != torch.int32:
            raise _lib.CfHipError("cf_serialize_nuscenes: sample tables must be int32")
        S = sample_ptr.numel() - 1
        order = torch.empty((S, max_per_sample), device=dev, dtype=torch.int32)
        counts = torch.empty((S,), device=dev, dtype=torch.int32)
        a.n_samples, a.sample_ptr, a.sample_frames = S, sample_ptr.data_ptr(), sample_frames.data_ptr()
        a.max_per_sample, a.order, a.counts = int(max_per_sample), order.data_ptr(), counts.data_ptr()
    _lib.check(_lib.load().cf_serialize_nuscenes(C.byref(a), _lib.stream_ptr()), "cf_serialize_nuscenes")
    return rows, rotation, order, counts


def pillar_expand(pc_2d, pc_3d, counts, calib, trans, out_hw, pillar_dims=(1.5, 0.2, 0.2),
                  want_aux=False):
    """pc_2d (B,3,Nmax) f64, pc_3d (B,R,Nmax) f64, counts (B) i32, calib (B,3,4) f64,
    trans (B,2,3) f64  ->  pc_dep (B,3,H,W) f32 [, keep (B,Nmax) u8, xy (B,2,Nmax) f64]."""
    _need_cuda(pc_2d, pc_3d, counts, calib, trans)
    B, _, max_n = pc_2d.shape
    H, W = out_hw
    dev = pc_2d.device
    pc_dep = torch.empty((B, 3, H, W), device=dev, dtype=torch.float32)
    keep = torch.empty((B, max_n), device=dev, dtype=torch.uint8) if want_aux else None
    xy = torch.empty((B, 2, max_n), device=dev, dtype=torch.float64) if want_aux else None
    for t, dt in ((pc_2d, torch.float64), (pc_3d, torch.float64), (counts, torch.int32),
                  (calib, torch.float64), (trans, torch.float64)):
        if t.dtype != dt or not t.is_contiguous():
            raise _lib.CfHipError("cf_pillar_expand: wrong dtype / non-contiguous input")
    h, w, l = pillar_dims
    _lib.check(_lib.load().cf_pillar_expand(pc_2d.data_ptr(), pc_3d.data_ptr(), counts.data_ptr(),
                                            B, max_n, pc_3d.shape[1], calib.data_ptr(),
                                            trans.data_ptr(), H, W, float(h), float(w), float(l),
                                            pc_dep.data_ptr(), _lib.ptr(keep), _lib.ptr(xy),
                                            _lib.stream_ptr()), "cf_pillar_expand")
    return (pc_dep, keep, xy) if want_aux else pc_dep


def stem_args(ps, x, out, shape=None, out_pool=None, in_scales=None) -> _lib.StemArgs:
    """x may be None with shape=(B, C, H, W) given: the image pointer is then patched in per call.
    out_pool: optional (B, H/4, W/4, 32) buffer for the 2x2 max-pool of the level1 map.
    in_scales: activation pre-scales (powers of two) of the image, base_layer's output, level0's output (None = 16 each)."""
    a = _lib.StemArgs()
    B, Cc, H, W = shape if x is None else x.shape
    a.x, a.B, a.C, a.H, a.W = (_lib.ptr(x), B, Cc, H, W)
    a.w_base, a.b_base, a.scale_base = ps.w_base.data_ptr(), ps.b_base.data_ptr(), ps.scale_base
    a.w_level0, a.b_level0, a.scale_level0 = ps.w_level0.data_ptr(), ps.b_level0.data_ptr(), ps.scale_level0
    a.w_level1, a.b_level1, a.scale_level1 = ps.w_level1.data_ptr(), ps.b_level1.data_ptr(), ps.scale_level1
    a.out = out.data_ptr()
    a.out_pool = _lib.ptr(out_pool)
    if in_scales is not None:
        for i, (s, name) in enumerate(zip(in_scales, ("scale_base", "scale_level0", "scale_level1"))):
            if s is not None and float(s) != DEFAULT_IN_SCALE:
                a.in_scale[i] = float(s)
                setattr(a, name, getattr(a, name) * DEFAULT_IN_SCALE / float(s))
    return a


def stem_fused(ps, x, out=None, out_pool=None):
    """images (B, C<=3, H, W) fp32 NCHW -> level1 map (B, H/2, W/2, 32) fp32 NHWC (packing.pack_stem); out_pool: also its
    MaxPool2d(2, 2), (B, H/4, W/4, 32)."""
    _need_cuda(x, out_pool)
    B, Cc, H, W = x.shape
    if out is None:
        out = torch.empty((B, H // 2, W // 2, 32), device=x.device, dtype=torch.float32)
    a = stem_args(ps, x.contiguous(), out, out_pool=out_pool)
    _lib.check(_lib.load().cf_stem_fused(C.byref(a), _lib.stream_ptr()), "cf_stem_fused")
    return out


def radar_ingest(pc, counts, intrinsics, img_wh, max_dist=60.0, z_offset=0.0, descending=False):
    """pc (B,R,Nmax) f64 padded raw sweeps, counts (B) i32, intrinsics (B,3,3) f64 ->
    pc_2d (B,3,Nmax) f64, pc_3d (B,R,Nmax) f64, counts_out (B) i32 - the inputs of pillar_expand."""
    _need_cuda(pc, counts, intrinsics)
    for t, dt in ((pc, torch.float64), (counts, torch.int32), (intrinsics, torch.float64)):
        if t.dtype != dt or not t.is_contiguous():
            raise _lib.CfHipError("cf_radar_ingest: wrong dtype / non-contiguous input")
    B, R, max_n = pc.shape
    pc_2d = torch.empty((B, 3, max_n), device=pc.device, dtype=torch.float64)
    pc_3d = torch.empty((B, R, max_n), device=pc.device, dtype=torch.float64)
    cnt = torch.empty((B,), device=pc.device, dtype=torch.int32)
    _lib.check(_lib.load().cf_radar_ingest(pc.data_ptr(), counts.data_ptr(), B, R, max_n, intrinsics.data_ptr(),
                                           int(img_wh[0]), int(img_wh[1]), float(max_dist), float(z_offset),
                                           int(bool(descending)), pc_2d.data_ptr(), pc_3d.data_ptr(), cnt.data_ptr(),
                                           _lib.stream_ptr()), "cf_radar_ingest")
    return pc_2d, pc_3d, cnt
