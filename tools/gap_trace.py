#!/usr/bin/env python
"""Where a map iteration's wall time goes: kernel time vs gaps, from a rocprofv3 --kernel-trace CSV.
    python tools/gap_trace.py <kernel_trace.csv> [anchor kernel substring = map_tail_rows] [skip iterations = 25]
Iterations are cut at the anchor kernel; prints mean busy / idle per iteration and the gaps (> 2 us) by the kernel that
FOLLOWS them."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(?:rtgs(?:_icp)?::)([A-Za-z0-9_]+?)(?:_kernel)?(<[^>]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name.split("(")[0][:40]


def main():
    path = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "map_tail_rows"
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    its, cur = [], []
    for r in rows:
        cur.append(r)
        if anchor in r[2]:
            its.append(cur)
            cur = []
    its = its[skip:]
    if not its:
        print("no iterations found")
        return
    busy = idle = wall = 0.0
    gaps = defaultdict(lambda: [0.0, 0])
    ktime = defaultdict(lambda: [0.0, 0])
    prev_end = None
    for it in its:
        for (s, e, k) in it:
            if prev_end is not None:
                g = max(0, s - prev_end)
                idle += g
                if g > 2000:
                    gaps[k][0] += g
                    gaps[k][1] += 1
            busy += e - max(s, prev_end or s)
            ktime[k][0] += e - s
            ktime[k][1] += 1
            prev_end = max(e, prev_end or e)
    n = len(its)
    wall = (its[-1][-1][1] - its[0][0][0] + 0.0) / 1e3
    print(f"{n} iterations: wall {wall / n:.1f} us/iter (first kernel start .. last kernel end), busy {busy / n / 1e3:.1f} us, idle {idle / n / 1e3:.1f} us")
    print("gaps > 2 us, by following kernel (us per iteration, count per iteration):")
    for k, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0]):
        print(f"  {k:28s} {g / n / 1e3:7.1f} us  x{c / n:.2f}")
    print("kernels (us per iteration):")
    for k, (t, c) in sorted(ktime.items(), key=lambda kv: -kv[1][0]):
        print(f"  {k:28s} {t / n / 1e3:7.1f} us  x{c / n:.2f}")


if __name__ == "__main__":
    main()
