"""Build profiles/<tag>_pmc_hbm_traffic.json from two rocprofv3 PMC passes of bench.py (separate runs:
FETCH_SIZE and WRITE_SIZE do not fit one pass):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r1_e_pmc_hbm_traffic.json [steps+warmup]
The file carries `_meta`: the sha256 of the kernel sources it was measured on (centerfusiondetect3d_amd.build.sources_sha -
bench.py reports `traffic` only while that matches the tree it runs from), the forward passes of the profiled command
and the HBM-side bytes of ONE pass over all kernels.
Units / corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: the counters are in
KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide streaming reads at 64 B, so it is
doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import os, glob, json, re, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centerfusiondetect3d_amd.build import sources_sha

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)

def per_kernel(d, counter):
    path = max(glob.glob(d + "/**/*.db", recursive=True), key=os.path.getmtime)
    c = sqlite3.connect(path)
    out = {}
    for name, total, n in c.execute("select kernel_name, sum(value), count(*) from counters_collection "
                                     "where counter_name = ? group by kernel_name", (counter,)):
        out[short(name)] = (total, n)
    return out

fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
res = {}
for k, (f, n) in sorted(fetch.items(), key=lambda kv: -kv[1][0]):
    w, nw = write.get(k, (0.0, 1))
    res[k] = {"launches": n, "fetch_bytes_per_launch_x2_corrected": f * 1024 * 2 / n,
              "write_bytes_per_launch": w * 1024 / max(nw, 1)}
passes = int(sys.argv[4]) if len(sys.argv) > 4 else None
total = sum((v["fetch_bytes_per_launch_x2_corrected"] + v["write_bytes_per_launch"]) * v["launches"] for v in res.values())
res["_meta"] = {"sources_sha": sources_sha(), "forward_passes_profiled": passes,
                "hbm_bytes_per_forward_all_kernels": (total / passes) if passes else None}
json.dump(res, open(sys.argv[3], "w"), indent=1)
del res["_meta"]
for k, v in list(res.items())[:14]:
    print(f"{k[:64]:64s} {v['launches']:5d}  fetch {v['fetch_bytes_per_launch_x2_corrected'] / 1e6:9.1f} MB  write {v['write_bytes_per_launch'] / 1e6:8.1f} MB")
