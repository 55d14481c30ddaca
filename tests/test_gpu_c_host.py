"""A host that is not Python: examples/c_host_conv.c (plain C, gcc) packs a BasicBlock convolution with cf_pack_conv_f16x3, runs it
through cf_conv3x3_f16x3 and checks it against a double-precision convolution + BatchNorm + ReLU on the CPU - the C ABI of
include/cf_hip.h used the way INTEGRATION.md describes, with no Python between the caller and libcfhip.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "centerfusiondetect3d_amd")


def _build(tmp_path):
    exe = str(tmp_path / "c_host_conv")
    subprocess.check_call(["gcc", "-O2", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                           os.path.join(ROOT, "examples", "c_host_conv.c"), "-o", exe, "-L" + PKG, "-lcfhip", "-L/opt/rocm/lib",
                           "-lamdhip64", "-lm", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_c_host_example_builds_as_plain_c(tmp_path):
    from centerfusiondetect3d_amd import build
    build.build(verbose=False)
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_c_host_example_runs(tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "max |err| / max |ref|" in r.stdout
