"""Dev tool: per-queue / per-stream view of the LAST step of a rocprofv3 --kernel-trace results .db: for each queue the time of
its first and last kernel, its busy time, and how long it ran ALONE (no kernel of any other queue active).
    python tools/prof_queues.py <dir or results.db>"""
import glob, os, sqlite3, sys
path = sys.argv[1]
if not path.endswith(".db"):
    path = max(glob.glob(path + "/**/*.db", recursive=True), key=os.path.getmtime)
c = sqlite3.connect(path)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = next((k for k in ("stream_id", "queue_id", "queue") if k in cols), None)
print("columns:", cols)
if qcol is None:
    sys.exit("no queue / stream column in the kernels view")
rows = list(c.execute(f"select name, start, end, {qcol} from kernels order by start"))
ends = [i for i, r in enumerate(rows) if "decode_post_kernel" in r[0] or "decode_gather_kernel" in r[0]]
last = rows[ends[-2] + 1:ends[-1] + 1]
t0 = last[0][1]
qs = {}
for n, s, e, q in last:
    d = qs.setdefault(q, [s, e, 0, 0])
    d[0], d[1], d[2], d[3] = min(d[0], s), max(d[1], e), d[2] + (e - s), d[3] + 1
for q, (s, e, busy, n) in sorted(qs.items(), key=lambda kv: kv[1][0]):
    # alone: parts of this queue's kernels during which no kernel of another queue runs
    others = sorted((a, b) for nn, a, b, qq in last if qq != q)
    alone = 0
    for nn, a, b, qq in last:
        if qq != q:
            continue
        cur = a
        for oa, ob in others:
            if ob <= cur or oa >= b:
                continue
            if oa > cur:
                alone += oa - cur
            cur = max(cur, ob)
            if cur >= b:
                break
        if cur < b:
            alone += b - cur
    print(f"{qcol} {q}: {n:4d} kernels, first start {(s - t0) / 1e3:8.1f} us, last end {(e - t0) / 1e3:8.1f} us, busy {busy / 1e3:8.1f} us, alone {alone / 1e3:8.1f} us")
