"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals and the per-launch timeline of the
LAST forward+decode step (launch order == plan order).
    python tools/prof_summary.py <kernel_trace.csv> [--timeline]
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a forward starts with the image layout kernel; a step = [marker, next marker)
    marks = [i for i, r in enumerate(rows) if "nchw_to_nhwc4" in r["Kernel_Name"]]
    per_step = marks[-1] - marks[-2]
    last = rows[marks[-2]:marks[-1]]
    span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e6
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / 1e6
    print(f"{len(rows)} launches in the trace, {per_step} per step; one steady-state step: span {span:.3f} ms, "
          f"kernel-busy {busy:.3f} ms")
    agg = defaultdict(lambda: [0, 0.0])
    for r in last:
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print(f"{'kernel':60s} {'calls':>5s} {'ms':>9s} {'%':>6s}")
    for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:60s} {n:5d} {ms:9.3f} {100 * ms / busy:6.1f}")
    if "--timeline" in sys.argv:
        t0 = int(last[0]["Start_Timestamp"])
        prev_end = t0
        for i, r in enumerate(last):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            print(f"{i:4d} {short(r['Kernel_Name']):50s} grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):6d} "
                  f"start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f} us")
            prev_end = e


if __name__ == "__main__":
    main()
