"""Build libcfhip.so (the C-ABI HIP library) in-tree for gfx950 with hipcc.

    python -m centerfusiondetect3d_amd.build            # incremental
    python -m centerfusiondetect3d_amd.build --force

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "_build")
LIB = os.path.join(PKG, "libcfhip.so")

ARCH = "gfx950"
COMMON = os.environ.get("CF_EXTRA_FLAGS", "").split() + ["-O3", "-std=c++17", "-fPIC", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}"]
# (source, extra flags).  cf_post: the index path must not fuse mul+add (bit-exact slice bounds).
SOURCES = [
    ("cf_gemm.hip", []),
    ("cf_gemm_bf16.hip", []),
    ("cf_gemm_f16.hip", []),
    ("cf_conv3x3_f16.hip", []),
    ("cf_stem.hip", []),
    ("cf_heads.hip", []),
    ("cf_elementwise.hip", []),
    ("cf_post.hip", ["-ffp-contract=off"]),
    ("cf_pack.cpp", ["-ffp-contract=off"]),     # host-side packers: the fp32 BN fold must round like torch's (no fused multiply-add)
    ("cf_error.cpp", []),
]


def sources_sha():
    """sha256 over every source the library is built from (csrc/*, include/*.h), by sorted relative path: what a
    committed profile was measured on (profiles/*_pmc_hbm_traffic.json carry it; bench.py reports `traffic` only when it
    matches the tree it runs from)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*")) + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libcfhip.so")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    # every header any source may include: editing one rebuilds everything that could depend on it
    headers = sorted(glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(CSRC, "*.h"))) + [__file__]
    objs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc, f"--offload-arch={ARCH}", *COMMON, *extra, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
