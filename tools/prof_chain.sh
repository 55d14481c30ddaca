#!/bin/bash
# Dev (GPU box): kernel durations of the index chain alone (rocprofv3 --stats over tools/bench_chain.py) -> gpurun_out/<tag>_chain_stats.txt
set -e
TAG=${1:-chain}; B=${2:-16}; shift || true; shift || true
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/chain_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -- python3 tools/bench_chain.py $B "$@" > $OUT.log 2>&1
python3 - "$OUT" > gpurun_out/${TAG}_chain_stats.txt <<'PY'
import sys, glob, os, sqlite3, re
from collections import defaultdict
path = max(glob.glob(sys.argv[1] + "/**/*.db", recursive=True), key=os.path.getmtime)
c = sqlite3.connect(path)
agg = defaultdict(list)
for n, s, e in c.execute("select name, start, end from kernels order by start"):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*$", "", n)
    agg[n].append((e - s) / 1e3)
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{n[:70]:70s} calls {len(v):5d}  median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}  max {v[-1]:8.1f}")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -delete; find $OUT -type d -empty -delete
cat gpurun_out/${TAG}_chain_stats.txt
