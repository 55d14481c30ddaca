"""CPU: libcfhip.so loads, exports every symbol include/cf_hip.h declares, and the ctypes
mirrors of the argument structs have the C compiler's layout.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cf_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cf_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported_and_bound():
    from centerfusiondetect3d_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), f"{n} declared in cf_hip.h but not exported by libcfhip.so"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes binding"
    assert set(_lib.SYMBOLS) == set(names)
    assert lib.cf_abi_version() == 6
    assert lib.cf_topk_workspace_bytes(16, 100) == 16 * 16 * 100 * 8
    assert lib.cf_topk_workspace_bytes_nms(16, 10, 112, 200, 100) == 16 * 16 * 100 * 8 + 16 * 10 * 112 * 200 * 4


def test_struct_layouts_match_c(tmp_path):
    from centerfusiondetect3d_amd import _lib
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "cf_hip.h"\nint main(){'
                    'printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(cf_conv_args), sizeof(cf_dcn_args),'
                    'sizeof(cf_decode_args), sizeof(cf_slot), offsetof(cf_conv_args, weight),'
                    'offsetof(cf_conv_args, precise), offsetof(cf_dcn_args, precise));return 0;}')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got == [ctypes.sizeof(_lib.ConvArgs), ctypes.sizeof(_lib.DcnArgs), ctypes.sizeof(_lib.DecodeArgs),
                   16, _lib.ConvArgs.weight.offset, _lib.ConvArgs.precise.offset, _lib.DcnArgs.precise.offset]
    # the structs added later: stem, head tail / fused head (size and the last field's offset pin the layout)
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "cf_hip.h"\nint main(){'
                    'printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(cf_stem_args), offsetof(cf_stem_args, out),'
                    'sizeof(cf_head_tail_args), offsetof(cf_head_tail_args, act),'
                    'sizeof(cf_head_fused_args), offsetof(cf_head_fused_args, first_scale),'
                    'offsetof(cf_dcn_args, workspace), offsetof(cf_dcn_args, out_split_bf16));'
                    'printf("%zu %zu %zu %zu\\n", offsetof(cf_conv_args, in_scale), offsetof(cf_dcn_args, in_scale),'
                    'offsetof(cf_dcn_args, mx_scale), offsetof(cf_stem_args, in_scale));return 0;}')
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got == [ctypes.sizeof(_lib.StemArgs), _lib.StemArgs.out.offset,
                   ctypes.sizeof(_lib.HeadTailArgs), _lib.HeadTailArgs.act.offset,
                   ctypes.sizeof(_lib.HeadFusedArgs), _lib.HeadFusedArgs.first_scale.offset,
                   _lib.DcnArgs.workspace.offset, _lib.DcnArgs.out_split_bf16.offset,
                   _lib.ConvArgs.in_scale.offset, _lib.DcnArgs.in_scale.offset, _lib.DcnArgs.mx_scale.offset,
                   _lib.StemArgs.in_scale.offset]           # (ABI 6: the activation pre-scales)


def test_serialize_struct_layout(tmp_path):
    from centerfusiondetect3d_amd import _lib
    prog = tmp_path / "sz2.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "cf_hip.h"\nint main(){'
                    'printf("%zu %zu %zu %zu\\n", sizeof(cf_serialize_args), offsetof(cf_serialize_args, rows),'
                    'offsetof(cf_serialize_args, max_per_sample), offsetof(cf_serialize_args, counts));return 0;}')
    exe = tmp_path / "sz2"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got == [ctypes.sizeof(_lib.SerializeArgs), _lib.SerializeArgs.rows.offset,
                   _lib.SerializeArgs.max_per_sample.offset, _lib.SerializeArgs.counts.offset]
    lib = _lib.load()
    assert lib.cf_serialize_max_candidates() >= 600
    a = _lib.SerializeArgs()
    assert lib.cf_serialize_nuscenes(ctypes.byref(a), None) == -22
    d = _lib.DecodeArgs()
    assert lib.cf_decode_post(ctypes.byref(d), None, None, None, None) == -22


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device."""
    from centerfusiondetect3d_amd import _lib
    lib = _lib.load()
    a = _lib.ConvArgs()
    assert lib.cf_conv2d_fused(ctypes.byref(a), None) == -22
    assert b"n_src" in lib.cf_last_error()
    d = _lib.DcnArgs()
    d.C = 48
    assert lib.cf_dcn_v2_fused(ctypes.byref(d), None) == -22
    assert b"multiple of 32" in lib.cf_last_error()
    assert lib.cf_topk_peaks(None, 1, 1, 1, 1, 1, 0, None, None, None, None, None) == -22
    assert lib.cf_upsample_dw(None, None, None, None, 1, 1, 1, 4, 2, None) == -22


def test_missing_library_fails_loudly(monkeypatch):
    from centerfusiondetect3d_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libcfhip.so")
    with pytest.raises(_lib.CfHipError, match="no CPU fallback"):
        _lib.load()


def test_integration_md_shows_the_real_dcn_args():
    """The ctypes struct INTEGRATION.md section 3b prints must be _lib.DcnArgs: same field names, types and size
    (the round-2 document showed a struct five fields short)."""
    import ctypes as C
    import re
    from centerfusiondetect3d_amd import _lib
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    m = re.search(r"^class DcnArgs\(C\.Structure\):.*?\n((?: {4}.*\n)+)", text, re.M)
    assert m, "INTEGRATION.md no longer shows class DcnArgs"
    ns = {"C": C}
    exec(m.group(0), ns)
    doc = ns["DcnArgs"]
    assert [(n, t) for n, t in doc._fields_] == [(n, t) for n, t in _lib.DcnArgs._fields_]
    assert C.sizeof(doc) == C.sizeof(_lib.DcnArgs)
    assert f"CF_ABI_VERSION {_lib.ABI_VERSION}" in m.group(0)


def test_graft_entry_build_runs():
    """The driver's build check (`__graft_entry__.build()`): incremental compile, dlopen, every header symbol resolved and
    the ABI version the binding was written against (it carried a stale literal once)."""
    import importlib, os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    try:
        importlib.import_module("__graft_entry__").build()
    finally:
        sys.path.remove(root)


def _bn(co, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.2, torch.randn(co, generator=g) * 0.3,
            torch.rand(co, generator=g) + 0.3)


def _c_bn(bn, keep):
    from centerfusiondetect3d_amd import _lib
    import ctypes as C
    b = _lib.PackBn()
    if bn is not None:
        arrs = [np.ascontiguousarray(t.numpy(), dtype=np.float32) for t in bn]
        keep += arrs
        b.gamma, b.beta, b.mean, b.var = (a.ctypes.data for a in arrs)
        b.eps = 1e-5
    return b


@pytest.mark.parametrize("co,srcs,k,stride,bias,bn,proj", [
    (64, [(64, 64, 0)], 3, 1, False, True, None),                 # BasicBlock conv: slice-major 3x3, BatchNorm folded
    (64, [(32, 32, 0)], 3, 2, False, True, None),                 # stride 2
    (27, [(64, 64, 0)], 3, 1, True, False, None),                 # conv_offset_mask: 27 -> N_pad 32, bias, no BatchNorm
    (64, [(64, 64, 0), (64, 64, 0)], 1, 1, False, True, None),    # Root: two sources
    (128, [(128, 128, 0), (128, 128, 0), (64, 64, 0), (128, 128, 0)], 1, 1, False, True, None),
    (128, [(128, 128, 0)], 3, 1, False, True, 64),                # conv2 + the Tree's project (cf_conv3x3_proj_f16x3)
    (64, [(24, 32, 8)], 3, 1, True, False, None),                 # not a multiple of 16 channels, offset into a wider tensor: generic slot order
])
def test_c_packer_equals_the_python_packer(co, srcs, k, stride, bias, bn, proj):
    """cf_pack_conv_f16x3 (host-side C, SURVEY 8(b) cf_pack_weights): BN fold + slot table + fp16 hi / lo fragments - the very bytes
    packing.fold_bn + packing.pack_conv_f16 produce, so a host in any language can feed the f16x3 convolution operators."""
    import ctypes as C
    from centerfusiondetect3d_amd import _lib, packing
    lib = _lib.load()
    g = torch.Generator().manual_seed(co * 7 + k)
    ci = sum(s[0] for s in srcs)
    w = torch.randn(co, ci, k, k, generator=g) * (ci * k * k) ** -0.5
    b = torch.randn(co, generator=g) if bias else None
    bnp = _bn(co, 1) if bn else None
    pw = pb_bn = None
    if proj:
        pw = torch.randn(co, proj, 1, 1, generator=g) * proj ** -0.5
        pb_bn = _bn(co, 2)
    # ---- Python
    wf, bf = packing.fold_bn(w, b, bnp)
    pyproj = None
    if proj:
        pwf, pbf = packing.fold_bn(pw, None, pb_bn)
        pyproj = (pwf, pbf, packing.Source(proj, proj))
    pc = packing.pack_conv_f16(wf, bf, [packing.Source(*s) for s in srcs], stride=stride, proj=pyproj)
    # ---- C
    keep = []
    d = _lib.PackConvDesc()
    wn = np.ascontiguousarray(w.numpy(), dtype=np.float32); keep.append(wn)
    d.weight = wn.ctypes.data
    if b is not None:
        bn_ = np.ascontiguousarray(b.numpy(), dtype=np.float32); keep.append(bn_)
        d.bias = bn_.ctypes.data
    d.bn = _c_bn(bnp, keep)
    d.cout, d.kh, d.kw, d.stride, d.pad, d.dilation = co, k, k, stride, -1, 1
    sa = (_lib.PackSrc * len(srcs))(*[_lib.PackSrc(*s) for s in srcs])
    d.src, d.n_src = sa, len(srcs)
    if proj:
        pwn = np.ascontiguousarray(pw.numpy().reshape(co, proj), dtype=np.float32); keep.append(pwn)
        d.proj_weight = pwn.ctypes.data
        d.proj_bn = _c_bn(pb_bn, keep)
        d.proj = _lib.PackSrc(proj, proj, 0)
    info = _lib.PackInfo()
    assert lib.cf_pack_conv_f16x3_info(C.byref(d), C.byref(info)) == 0, lib.cf_last_error()
    assert (info.n_pad, info.k_pad, bool(info.patch), info.n_slots) == (pc.n_pad, pc.k_pad, bool(pc.patch), pc.slots.shape[0])
    wout = np.zeros(info.weight_bytes // 2, np.uint16)
    sout = np.zeros((info.n_slots, 4), np.int32)
    bout = np.zeros(info.n_pad, np.float32)
    assert lib.cf_pack_conv_f16x3(C.byref(d), wout.ctypes.data, sout.ctypes.data, bout.ctypes.data, C.byref(info)) == 0, lib.cf_last_error()
    assert np.array_equal(sout, pc.slots.numpy())
    assert np.array_equal(bout, pc.bias.numpy())
    assert info.out_scale == pc.out_scale
    assert np.array_equal(wout, pc.weight.contiguous().view(torch.int16).numpy().view(np.uint16).ravel())


@pytest.mark.parametrize("co,ci", [(64, 64), (64, 128), (256, 512), (40, 32)])
def test_c_dcn_packer_equals_the_python_packer(co, ci):
    import ctypes as C
    from centerfusiondetect3d_amd import _lib, packing
    lib = _lib.load()
    g = torch.Generator().manual_seed(co + ci)
    w, b, bnp = torch.randn(co, ci, 3, 3, generator=g) * (ci * 9) ** -0.5, torch.randn(co, generator=g), _bn(co, 3)
    wf, bf = packing.fold_bn(w, b, bnp)
    pd = packing.pack_dcn_f16(wf, bf)
    keep = []
    wn, bn_ = np.ascontiguousarray(w.numpy(), dtype=np.float32), np.ascontiguousarray(b.numpy(), dtype=np.float32)
    cbn = _c_bn(bnp, keep)
    info = _lib.PackInfo()
    assert lib.cf_pack_dcn_f16_info(co, ci, C.byref(info)) == 0
    wout, bout = np.zeros(info.weight_bytes // 2, np.uint16), np.zeros(info.n_pad, np.float32)
    assert lib.cf_pack_dcn_f16(wn.ctypes.data, bn_.ctypes.data, C.byref(cbn), co, ci, wout.ctypes.data, bout.ctypes.data, C.byref(info)) == 0
    assert info.n_pad == pd.n_pad and info.out_scale == pd.out_scale
    assert np.array_equal(bout, pd.bias.numpy())
    assert np.array_equal(wout, pd.weight.contiguous().view(torch.int16).numpy().view(np.uint16).ravel())
    # a descriptor the packer cannot honour is refused with a message, not packed wrongly
    assert lib.cf_pack_dcn_f16_info(co, 20, C.byref(info)) != 0 and b"cin" in lib.cf_last_error()
