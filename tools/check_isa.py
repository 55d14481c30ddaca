"""ISA invariant behind DESIGN.md section 6 ("the 4e03164 glitch"): inside the scheduling-pinned main loops
(`__builtin_amdgcn_sched_barrier` regions) of the MFMA kernels there must be NO exec-masked vector-memory load -
every lane always loads from a valid (clamped) address and out-of-range contributions are removed by a select or a
zero weight.  (Operand-class loads = dwordx2/x4; the scalar-width bias reads of the store epilogues are consumed at once.)  Compiles the sources to gfx950 assembly (no GPU needed) and scans them.
    python tools/check_isa.py            -> exit status 0 if the invariant holds"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "centerfusiondetect3d_amd", "csrc")
SOURCES = {"cf_conv3x3_f16.hip": ("conv3x3_f16x3_kernel",), "cf_heads.hip": ("head_patch_kernel", "head_patch16_kernel"),
           "cf_gemm_f16.hip": ("dcn_f16x3_kernel",)}


def functions(asm):
    cur, out = None, {}
    for ln in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif ln.startswith(".Lfunc_end"):
            cur = None
        elif cur:
            out[cur].append(ln)
    return out


def violations(lines):
    """exec-masked vector loads between the first and the last sched_barrier marker of a function."""
    marks = [i for i, l in enumerate(lines) if "sched_barrier" in l]
    if not marks:
        return None
    bad, masked = [], False
    for i in range(marks[0], marks[-1] + 1):
        l = lines[i]
        if "s_and_saveexec" in l:
            masked = True
        elif re.search(r"s_or_b64\s+exec", l) or re.search(r"s_mov_b64\s+exec", l) or "s_endpgm" in l or re.search(r"\bs_branch\b", l):
            masked = False                    # (straight-line tracking: a return or an unconditional branch ends the masked fall-through)
        elif masked and re.search(r"\b(global|buffer|flat)_load_dwordx[234]", l):     # operand-class (prefetch) loads
            bad.append((i, l.strip()))
    return bad


def main():
    hipcc = "/opt/rocm/bin/hipcc"
    failed = False
    with tempfile.TemporaryDirectory() as tmp:
        for src, kernels in SOURCES.items():
            out = os.path.join(tmp, src + ".s")
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{os.path.join(ROOT, 'include')}",
                                   f"-I{CSRC}", "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, src)],
                                  stderr=subprocess.DEVNULL)
            for name, lines in functions(open(out).read()).items():
                if not any(k in name for k in kernels):
                    continue
                v = violations(lines)
                if v is None:
                    print(f"{src}: {name[:80]}: no pinned region")
                    continue
                print(f"{src}: {name[:90]}: {len(v)} exec-masked loads inside the pinned region")
                for i, l in v[:5]:
                    print(f"      line {i}: {l}")
                failed |= bool(v)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
