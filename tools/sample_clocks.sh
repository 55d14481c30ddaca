#!/bin/bash
# Dev (GPU box): shader clock and socket power while a command runs (rocm-smi polled every 0.25 s in the background).
#   bash tools/sample_clocks.sh <label> <command ...>   -> appends to gpurun_out/clock_samples.txt
LABEL=$1; shift
OUT=gpurun_out/clock_samples.txt
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power" | tr '\n' ' '; echo; sleep 0.25; done ) > /tmp/clk_$$.txt &
POLL=$!
"$@" > /tmp/cmd_$$.txt 2>&1
kill $POLL 2>/dev/null
echo "== $LABEL" >> $OUT
grep -E "tails|us " /tmp/cmd_$$.txt | head -5 >> $OUT
python3 - <<PY >> $OUT
import re
sclk, pw = [], []
for line in open("/tmp/clk_$$.txt"):
    m = re.search(r"sclk clock level: \S+: \((\d+)Mhz\)", line)
    if m: sclk.append(int(m.group(1)))
    m = re.search(r"Power \(W\): ([\d.]+)", line)
    if m: pw.append(float(m.group(1)))
import statistics as st
if sclk: print(f"  sclk MHz: n={len(sclk)} min {min(sclk)} median {st.median(sclk)} max {max(sclk)}")
if pw: print(f"  power W: min {min(pw):.0f} median {st.median(pw):.0f} max {max(pw):.0f}")
PY
tail -4 $OUT
