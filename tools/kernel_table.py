#!/usr/bin/env python
"""Per-kernel averages of a rocprofv3 --kernel-trace --stats run, short names:  python tools/kernel_table.py <dir> [top]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
print("%-50s %7s %10s %10s" % ("kernel", "calls", "avg us", "total ms"))
for r in rows[:top]:
    n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("rtgs::", "")[:50]
    print("%-50s %7s %10.1f %10.2f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
print("%-50s %7d %10s %10.2f   (all %d kernels)" % ("TOTAL", sum(int(r["Calls"]) for r in rows), "", tot, len(rows)))
