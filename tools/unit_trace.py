#!/usr/bin/env python
"""How the two streams of bench.py's unit (tracker || mapper) share the GPU, from a rocprofv3 --kernel-trace CSV of a
bench run:   python tools/unit_trace.py <kernel_trace.csv> [units to skip = 200] [units to take = 200]
Units are cut at map_fused_tail; kernels are split into the tracker's (rtgs_icp::) and the mapper's.  Prints per unit:
wall, the time either / both / neither stream has a kernel running, each side's kernel time, and the per-kernel average
duration (compare with the same kernels running alone: profiles/*_icp_kernel_stats.csv, *_headline_kernel_table.txt)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(?:rtgs(?:_icp)?::)([A-Za-z0-9_]+?)(?:_kernel)?(<[^>]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name.split("(")[0][:40]


def union(iv):
    iv = sorted(iv)
    out, tot = [], 0
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out, sum(e - s for s, e in out)


def overlap(a, b):
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if e > s:
            tot += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main():
    path = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    take = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), "rtgs_icp::" in r["Kernel_Name"]))
    rows.sort()
    ends = [r[1] for r in rows if "map_fused_tail" in r[2]]
    if len(ends) < skip + take + 1:
        print("not enough units:", len(ends))
        return
    t0, t1 = ends[skip], ends[skip + take]
    sel = [r for r in rows if r[0] >= t0 and r[1] <= t1]
    icp, _ = union([(s, e) for s, e, k, t in sel if t])
    mp, _ = union([(s, e) for s, e, k, t in sel if not t])
    both = overlap(icp, mp)
    ti, tm = sum(e - s for s, e in icp), sum(e - s for s, e in mp)
    wall = t1 - t0
    n = take
    print(f"{n} units: wall {wall / n / 1e3:.1f} us/unit | tracker kernels running {ti / n / 1e3:.1f} us, mapper kernels running {tm / n / 1e3:.1f} us, "
          f"both {both / n / 1e3:.1f} us, neither {(wall - ti - tm + both) / n / 1e3:.1f} us")
    kt = defaultdict(lambda: [0, 0])
    for s, e, k, t in sel:
        kt[("T " if t else "M ") + k][0] += e - s
        kt[("T " if t else "M ") + k][1] += 1
    print("kernel                          us/unit   calls/unit   avg us")
    for k, (t, c) in sorted(kt.items(), key=lambda kv: -kv[1][0])[:30]:
        print(f"  {k:30s} {t / n / 1e3:7.1f} {c / n:9.2f} {t / c / 1e3:10.1f}")


main()
