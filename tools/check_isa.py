"""ISA invariant behind DESIGN.md section 6 ("the 4e03164 glitch"): inside the scheduling-pinned main loops
(`__builtin_amdgcn_sched_barrier` regions) of the MFMA kernels there must be NO exec-masked vector-memory load -
every lane always loads from a valid (clamped) address and out-of-range contributions are removed by a select or a
zero weight.  (Operand-class loads = dwordx2/x4; the scalar-width bias reads of the store epilogues are consumed at once.)  Compiles the sources to gfx950 assembly (no GPU needed) and scans them.
Two more invariants on the same listings:
  * coalesced epilogues (conv_f16x3 / dcn_f16x3 / conv3x3_f16x3): the lanes of a wave exchange their output tile through
    LDS; between the tile's ds_write group and the first ds_read behind it there must be an `s_waitcnt lgkmcnt(0)`
    (cf_wave_lds_sync) - checked behind the last v_mfma of every such kernel;
  * no scratch at all (compiler-reported ScratchSize 0) in any instantiation of the kernels in NO_SCRATCH, and no
    `scratch_` instruction inside an MFMA stream of any scanned kernel (spills in the MFMA loops);
    `--scratch-report` only lists them.
    python tools/check_isa.py            -> exit status 0 if the invariants hold"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "centerfusiondetect3d_amd", "csrc")
SOURCES = {"cf_conv3x3_f16.hip": ("conv3x3_f16x3_kernel",), "cf_heads.hip": ("head_patch_kernel", "head_patch16_kernel"),
           "cf_gemm_f16.hip": ("dcn_f16x3_kernel", "conv_f16x3_kernel")}


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


EPILOGUE_KERNELS = ("conv_f16x3_kernel", "dcn_f16x3_kernel", "conv3x3_f16x3_kernel")
# kernels whose MFMA loop must be free of scratch traffic (every instantiation the default path launches)
# kernels of the default path: NO instantiation may use scratch at all (ScratchSize 0 in the compiler's resource summary)
NO_SCRATCH = ("head_patch16_kernel", "head_patch_kernel", "dcn_f16x3_kernel", "conv3x3_f16x3_kernel", "conv_f16x3_kernel")


def scratch_sizes(asm):
    """{mangled kernel name: ScratchSize} from the '; ScratchSize: N' lines the compiler prints behind each kernel."""
    cur, out = None, {}
    for ln in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
        m = re.match(r"; ScratchSize: (\d+)", ln)
        if m and cur:
            out[cur] = int(m.group(1))
    return out


def epilogue_violations(lines):
    """ds_read behind a ds_write of the same wave with no `s_waitcnt lgkmcnt(0)` in between, in the epilogue = behind the
    `; cf_epilogue_begin` marker the kernels emit (asm volatile comment) where the tile exchange starts."""
    marks = [i for i, l in enumerate(lines) if "cf_epilogue_begin" in l]
    if not marks:
        return None                            # this instantiation stores its accumulators directly (no exchange)
    start = marks[0]
    bad, pending = [], False
    for i in range(start, len(lines)):
        l = lines[i].split(";")[0]
        if re.search(r"\bds_write", l):
            pending = True
        elif re.search(r"s_waitcnt.*lgkmcnt\(0\)", l) or re.search(r"\bs_barrier\b", l):
            pending = False
        elif re.search(r"\bs_branch\b", l) or "s_endpgm" in l:
            pending = False                    # (straight-line tracking, as in violations(): the compiler may lay a block of the main
                                               #  loop - the PROJ staging stores, say - out behind the epilogue marker; its writes end
                                               #  in a jump back and are not followed by the next block's reads)
        elif pending and re.search(r"\bds_read", l):
            bad.append((i, l.strip()))
            pending = False
    return bad


def scratch_in_mfma_loop(lines):
    """scratch_load / scratch_store inside an MFMA stream: in ONE basic block (no label, no branch in between) with a
    v_mfma before it and a v_mfma after it.  Spills at the boundaries of an outer loop (between blocks) do not count."""
    def edge(l):
        c = l.split(";")[0].strip()
        return bool(re.match(r"^\.?L?\w+:", c)) or bool(re.match(r"s_(c)?branch", c)) or c.startswith("s_endpgm")
    bad = []
    for i, l in enumerate(lines):
        if not re.match(r"\s*scratch_", l):
            continue
        before = after = False
        for j in range(i - 1, -1, -1):
            if edge(lines[j]):
                break
            if "v_mfma" in lines[j]:
                before = True
                break
        for j in range(i + 1, len(lines)):
            if edge(lines[j]):
                break
            if "v_mfma" in lines[j]:
                after = True
                break
        if before and after:
            bad.append((i, l.strip()))
    return bad


def main():
    hipcc = "/opt/rocm/bin/hipcc"
    failed = False
    report_only = "--scratch-report" in sys.argv
    seen_epilogues = set()
    with tempfile.TemporaryDirectory() as tmp:
        for src, kernels in SOURCES.items():
            out = os.path.join(tmp, src + ".s")
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{os.path.join(ROOT, 'include')}",
                                   f"-I{CSRC}", "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, src)],
                                  stderr=subprocess.DEVNULL)
            asm = open(out).read()
            for name, size in scratch_sizes(asm).items():
                if size and any(k in name for k in NO_SCRATCH):
                    print(f"{src}: {name[:90]}: ScratchSize {size} (spills)")
                    failed |= not report_only
            for name, lines in functions(asm).items():
                if not any(k in name for k in kernels):
                    continue
                if any(k in name for k in EPILOGUE_KERNELS):
                    ev = epilogue_violations(lines)
                    if ev is not None:
                        seen_epilogues.add(next(k for k in EPILOGUE_KERNELS if k in name))
                    if ev:
                        print(f"{src}: {name[:90]}: {len(ev)} epilogue ds_read not ordered behind its ds_write group")
                        failed = True
                sc = scratch_in_mfma_loop(lines)
                if sc:
                    print(f"{src}: {name[:90]}: {len(sc)} scratch ops inside an MFMA stream (same basic block, MFMAs on both sides)")
                    failed |= not report_only
                v = violations(lines)
                if v is None:
                    print(f"{src}: {name[:80]}: no pinned region")
                    continue
                print(f"{src}: {name[:90]}: {len(v)} exec-masked loads inside the pinned region")
                for i, l in v[:5]:
                    print(f"      line {i}: {l}")
                failed |= bool(v)
    missing = set(EPILOGUE_KERNELS) - seen_epilogues
    if missing:
        print(f"no instantiation of {sorted(missing)} carries the cf_epilogue_begin marker")
        failed = True
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
