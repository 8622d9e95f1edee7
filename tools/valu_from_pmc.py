#!/usr/bin/env python
"""profiles/valu_latest.json from a tools/pmc_sq.py table: per kernel the VALU wave-instructions per launch and how much
of the kernel's busy time the VALU was active, stamped with the hash of the kernel sources (bench.py's `roofline.valu`):
    python tools/valu_from_pmc.py <pmc_sq_*.csv> [more.csv ...] <out.json>
Later tables override earlier ones per kernel (give the scene whose kernels bench.py's headline leg runs LAST)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash
from pmc_summary import load


def main():
    out = {}
    for path in sys.argv[1:-1]:
        for k, c in load(path).items():
            if "SQ_INSTS_VALU" not in c:
                continue
            e = {"SQ_INSTS_VALU": int(c["SQ_INSTS_VALU"]), "table": os.path.basename(path)}
            if "SQ_BUSY_CYCLES" in c:
                # instructions per busy SIMD-cycle (SQ_BUSY_CYCLES is summed over the 32 SEs); the issue-rate probe's floor is
                # ~0.72 (one per 1.4 cycles).  Round 4's "valu_active_frac" multiplied the instruction count by a nominal 4
                # cycles - a definition, dropped.
                e["valu_per_busy_simd_cycle"] = round((c["SQ_INSTS_VALU"] / 1024) / (c["SQ_BUSY_CYCLES"] / 32), 4)
            out[k] = e
    out["_source_sha16"] = source_hash()
    json.dump(out, open(sys.argv[-1], "w"), indent=1)
    print(f"{len(out) - 1} kernels -> {sys.argv[-1]} (sources {out['_source_sha16']})")


if __name__ == "__main__":
    main()
