"""Per-kernel averages of every counter found in one or more rocprofv3 --pmc result directories
(one directory per pass; counters of different passes are joined on the kernel name).
    rocprofv3 --pmc A B C --kernel-trace -d gpurun_out/pmcX -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline
    python tools/pmc_kernels.py gpurun_out/pmcX [gpurun_out/pmcY ...] [--top 20]
Derived columns (when their inputs are present):
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)
  wait_any   = SQ_WAIT_ANY / SQ_WAVE_CYCLES         (wave parked in s_waitcnt / barrier)
  wait_inst  = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES    (issue stall)
  active     = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  lds_confl  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"""
import glob, os, re, sqlite3, sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)


def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 24
    data = defaultdict(dict)          # kernel -> counter -> mean per launch
    launches = {}
    for d in dirs:
        for path in glob.glob(d + "/**/*.db", recursive=True):
            c = sqlite3.connect(path)
            try:
                rows = list(c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                                      "group by kernel_name, counter_name"))
            except sqlite3.Error:
                continue
            for k, cn, tot, n in rows:
                data[short(k)][cn] = tot / n
                launches[short(k)] = n
    names = sorted({c for v in data.values() for c in v})
    print("counters:", " ".join(names))
    key = "SQ_BUSY_CU_CYCLES" if any("SQ_BUSY_CU_CYCLES" in v for v in data.values()) else (names[0] if names else "")
    order = sorted(data, key=lambda k: -data[k].get(key, 0) * launches.get(k, 1))[:top]
    for k in order:
        v = data[k]
        parts = []
        g = v.get
        if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("SQ_BUSY_CU_CYCLES"):
            parts.append(f"mfma_busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / (4 * g('SQ_BUSY_CU_CYCLES')):.3f}")
        if g("SQ_WAVE_CYCLES"):
            for lab, cn in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst", "SQ_WAIT_INST_ANY"), ("active", "SQ_ACTIVE_INST_ANY"),
                            ("a_valu", "SQ_ACTIVE_INST_VALU"), ("a_lds", "SQ_ACTIVE_INST_LDS"), ("a_vmem", "SQ_ACTIVE_INST_VMEM"),
                            ("a_sca", "SQ_ACTIVE_INST_SCA"), ("a_misc", "SQ_ACTIVE_INST_MISC"), ("w_lds", "SQ_WAIT_INST_LDS")):
                if g(cn) is not None:
                    parts.append(f"{lab} {g(cn) / g('SQ_WAVE_CYCLES'):.3f}")
        if g("SQ_LDS_IDX_ACTIVE"):
            parts.append(f"lds_confl {g('SQ_LDS_BANK_CONFLICT', 0) / g('SQ_LDS_IDX_ACTIVE'):.3f}")
        if g("SQ_INSTS_MFMA"):
            for lab, cn in (("valu/mfma", "SQ_INSTS_VALU"), ("lds/mfma", "SQ_INSTS_LDS"), ("vmem_rd/mfma", "SQ_INSTS_VMEM_RD"),
                            ("salu/mfma", "SQ_INSTS_SALU")):
                if g(cn) is not None:
                    x = g(cn) - (g("SQ_INSTS_MFMA") if cn == "SQ_INSTS_VALU" else 0)
                    parts.append(f"{lab} {x / g('SQ_INSTS_MFMA'):.2f}")
        print(f"{k[:62]:62s} n={launches[k]:4d}  " + "  ".join(parts))
        if "--raw" in sys.argv:
            print("     " + "  ".join(f"{c}={v[c]:.4g}" for c in names if c in v))


if __name__ == "__main__":
    main()
