"""CPU: libcfhip.so loads, exports every symbol include/cf_hip.h declares, and the ctypes
mirrors of the argument structs have the C compiler's layout.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

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
    assert lib.cf_abi_version() == 5
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
                    'offsetof(cf_dcn_args, workspace), offsetof(cf_dcn_args, out_split_bf16));return 0;}')
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got == [ctypes.sizeof(_lib.StemArgs), _lib.StemArgs.out.offset,
                   ctypes.sizeof(_lib.HeadTailArgs), _lib.HeadTailArgs.act.offset,
                   ctypes.sizeof(_lib.HeadFusedArgs), _lib.HeadFusedArgs.first_scale.offset,
                   _lib.DcnArgs.workspace.offset, _lib.DcnArgs.out_split_bf16.offset]


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
