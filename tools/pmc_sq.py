#!/usr/bin/env python
"""Per-kernel, per-launch means of rocprofv3 PMC passes:
    python tools/pmc_sq.py out.csv label1=<counter_collection.csv> [label2=<...> ...]
Counter values of one dispatch are summed over the dimensions rocprofv3 reports (XCC / SE / ...); launches that exit at
once (< 2 % of the kernel's largest launch in SQ_WAVE_CYCLES, when collected) are left out of the mean."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(?:rtgs(?:_icp)?::)([A-Za-z0-9_]+?)(?:_kernel)?(<[^>]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name.split("(")[0][:60]


def reduce(path):
    disp = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(path)):
        disp[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    big = defaultdict(float)
    for (k, _), c in disp.items():
        big[k] = max(big[k], c.get("SQ_WAVE_CYCLES", c.get("SQ_BUSY_CYCLES", 1.0)))
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    for (k, _), c in disp.items():
        if c.get("SQ_WAVE_CYCLES", c.get("SQ_BUSY_CYCLES", 1.0)) < 0.02 * big[k]:
            continue
        n[k] += 1
        for cn, v in c.items():
            acc[k][cn] += v
    return {short(k): ({cn: v / n[k] for cn, v in c.items()}, n[k]) for k, c in acc.items()}


def main():
    rows = ["pass,kernel,launches,counter,value_per_launch"]
    for arg in sys.argv[2:]:
        label, path = arg.split("=", 1)
        for k, (c, n) in sorted(reduce(path).items()):
            for cn, v in sorted(c.items()):
                rows.append(f"{label},{k},{n},{cn},{v:.0f}")
    open(sys.argv[1], "w").write("\n".join(rows) + "\n")
    print(f"{len(rows) - 1} rows -> {sys.argv[1]}")


if __name__ == "__main__":
    main()
