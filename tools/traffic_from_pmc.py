#!/usr/bin/env python
"""HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in separate runs as
/opt/skills/guides/MI355X_MICROARCH.md prescribes):
    python tools/traffic_from_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 - FETCH_SIZE is doubled (gfx950 reports half of 16-B/lane streaming
reads: exact for the streaming kernels, an upper bound for gather kernels); WRITE_SIZE is uncalibrated."""
import csv
import json
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    per_dispatch = defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        per_dispatch[(r["Kernel_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    biggest = defaultdict(float)
    for (name, _), v in per_dispatch.items():
        biggest[name] = max(biggest[name], v)
    for (name, _), v in per_dispatch.items():
        if v < 0.02 * biggest[name]:        # launches that exit at once (second blend_bwd / bin_count of the two-pass forward)
            continue
        acc[name][0] += v
        acc[name][1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


def short(name):
    m = re.search(r"(?:rtgs(?:_icp)?::)([A-Za-z0-9_]+?)(?:_kernel)?(<[^>]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else None


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for name, f in fetch.items():
        s = short(name)
        if not s:
            continue
        w = write.get(name, 0.0)
        out[s] = {"rocprof_kernel": name.split("(")[0], "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                  "hbm_bytes_per_launch": int((2 * f + w) * 1024),
                  "note": "(2*FETCH_SIZE + WRITE_SIZE)*1024, mean over the launches of the run"}
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import source_hash
    out["_source_sha16"] = source_hash()       # bench.py quotes these numbers only for the kernel sources they were measured on
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(((k, v) for k, v in out.items() if isinstance(v, dict)), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
        print(f"{k:40s} {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch")


if __name__ == "__main__":
    main()
