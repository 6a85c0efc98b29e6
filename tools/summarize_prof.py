#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats and PMC rows) into a small text summary that is
committed under profiles/.  Usage: summarize_prof.py <rocprof output dir> <out.txt>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(src, dst):
    lines = []
    for f in sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)):
        lines.append("# %s" % os.path.relpath(f, src))
        with open(f) as fh:
            rows = list(csv.DictReader(fh))
        lines.append("%-60s %8s %14s %14s %8s" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
        for r in rows:
            lines.append("%-60s %8s %14.3f %14.3f %8s" % (
                r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                float(r["AverageNs"]) / 1e6, r["Percentage"]))
    pmc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                pmc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if pmc:
        lines.append("# PMC counters: per-dispatch mean (n dispatches)")
        for k in sorted(pmc):
            for c in sorted(pmc[k]):
                v = pmc[k][c]
                lines.append("%-60s %-22s %18.1f  (n=%d)" % (k, c, sum(v) / len(v), len(v)))
    with open(dst, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
