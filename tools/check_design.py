#!/usr/bin/env python3
"""DESIGN.md stays a document: no table cell over 400 characters, no unfilled placeholder, and the test counts it
quotes are the ones `pytest --collect-only` reports.
    python tools/check_design.py [--counts | --fix-counts]      (--fix-counts rewrites the two numbers in section 6)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect_counts():
    out = {}
    for tag, expr in (("gpu", "gpu"), ("cpu", "not gpu")):
        r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-q", "--collect-only", "-m", expr],
                           cwd=ROOT, capture_output=True, text=True)
        m = re.search(r"(\d+)/(\d+) tests collected|(\d+) tests? collected", r.stdout)
        out[tag] = int(m.group(1) or m.group(3)) if m else None
    return out


def main():
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    bad = []
    for i, line in enumerate(text.split("\n"), 1):
        if line.startswith("|"):
            for cell in line.strip().strip("|").split("|"):
                if len(cell.strip()) > 400:
                    bad.append(f"DESIGN.md:{i}: table cell of {len(cell.strip())} characters")
    for m in re.finditer(r"\bR4_[A-Z_]+\b", text):
        bad.append(f"unfilled placeholder {m.group(0)}")
    if "--counts" in sys.argv or "--fix-counts" in sys.argv:
        c = collect_counts()
        print(f"collected: {c['gpu']} GPU tests, {c['cpu']} CPU tests")
        pat = r"(`-m gpu`: \*\*)(\d+)(\*\* tests.*?`-m \"not gpu\"`: \*\*)(\d+)(\*\* tests)"
        if "--fix-counts" in sys.argv and re.search(pat, text, re.S):
            text = re.sub(pat, lambda m: f"{m.group(1)}{c['gpu']}{m.group(3)}{c['cpu']}{m.group(5)}", text, count=1, flags=re.S)
            open(os.path.join(ROOT, "DESIGN.md"), "w").write(text)
        m = re.search(r"`-m gpu`: \*\*(\d+)\*\* tests.*?`-m \"not gpu\"`: \*\*(\d+)\*\* tests", text, re.S)
        if not m:
            bad.append("DESIGN.md section 6 does not quote the test counts in the expected form")
        elif (int(m.group(1)), int(m.group(2))) != (c["gpu"], c["cpu"]):
            bad.append(f"DESIGN.md quotes {m.group(1)} / {m.group(2)} tests, pytest collects {c['gpu']} / {c['cpu']}")
    for b in bad:
        print(b)
    print("DESIGN.md ok" if not bad else f"{len(bad)} problem(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
