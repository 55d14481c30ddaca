"""Summarise a rocprofv3 --kernel-trace results .db (sqlite): per-kernel totals of the LAST step of
bench.py (a step starts at the first kernel of the forward: the stem) and optionally its timeline.
    python tools/prof_db.py <results.db> [--timeline]
"""
import os, glob, re, sqlite3, sys
from collections import defaultdict

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)

path = sys.argv[1]
if not path.endswith(".db"):
    path = max(glob.glob(path + "/**/*.db", recursive=True), key=os.path.getmtime)
c = sqlite3.connect(path)
rows = list(c.execute("select name, start, end from kernels order by start"))
# a step ENDS with the decode kernel (one per step on every stream configuration; with model.streams > 1 a step holds
# one stem launch per trunk stream, so the stem cannot delimit it)
enders = ("decode_post_kernel", "decode_gather_kernel")
ends = [i for i, r in enumerate(rows) if any(e in r[0] for e in enders)]
if len(ends) >= 2:
    last = rows[ends[-2] + 1:ends[-1] + 1]
else:
    first = next(k for k in ("stem_kernel", "nchw_to_nhwc4") if any(k in r[0] for r in rows))
    marks = [i for i, r in enumerate(rows) if first in r[0]]
    last = rows[marks[-2]:marks[-1]]
span = (last[-1][2] - last[0][1]) / 1e6
busy = sum(e - s for _, s, e in last) / 1e6
print(f"{len(rows)} launches in the trace, {len(last)} per step; one steady-state step: span {span:.3f} ms, kernel-busy {busy:.3f} ms")
agg = defaultdict(lambda: [0, 0.0])
for n, s, e in last:
    agg[short(n)][0] += 1
    agg[short(n)][1] += (e - s) / 1e6
print(f"{'kernel':84s} {'calls':>5s} {'ms':>9s} {'%':>6s}")
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[:84]:84s} {n:5d} {ms:9.3f} {100 * ms / busy:6.1f}")
if "--timeline" in sys.argv:
    t0 = last[0][1]
    for n, s, e in last:
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  {short(n)[:90]}")
