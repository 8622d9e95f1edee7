#!/usr/bin/env python
"""One line per kernel from a tools/pmc_sq.py table: python tools/pmc_summary.py <pmc_sq_*.csv> [kernel ...]"""
import sys
from collections import defaultdict


def load(path):
    d = defaultdict(dict)
    for ln in open(path).read().splitlines()[1:]:
        f = ln.split(",")
        d[",".join(f[1:-3])][f[-2]] = float(f[-1])
    return d


def main():
    d = load(sys.argv[1])
    for k in (sys.argv[2:] or sorted(d)):
        c = d[k]
        if not c.get("SQ_INSTS_VALU") or not c.get("SQ_ACTIVE_INST_VALU") or "SQ_THREAD_CYCLES_VALU" not in c:
            continue
        # SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md); 1024 SIMDs, 32 SEs
        act = c["SQ_ACTIVE_INST_VALU"] * 4 / 1024
        print(f"{k:22s} VALU {c['SQ_INSTS_VALU'] / 1e6:7.2f}M SALU {c['SQ_INSTS_SALU'] / 1e6:6.2f}M LDS {c['SQ_INSTS_LDS'] / 1e6:6.2f}M | "
              f"busy {c['SQ_BUSY_CYCLES'] / 32 / 1e3:6.0f}k cyc/SE, VALU-active {act / 1e3:6.0f}k cyc/SIMD ({act / (c['SQ_BUSY_CYCLES'] / 32):.2f}), "
              f"{c['SQ_ACTIVE_INST_VALU'] * 4 / c['SQ_INSTS_VALU']:.2f} cyc/VALU, lanes lit {c['SQ_THREAD_CYCLES_VALU'] / (c['SQ_ACTIVE_INST_VALU'] * 4 * 64):.2f}, "
              f"wait_inst {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.2f} wait_any {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.2f} "
              f"waves {c['SQ_WAVES']:.0f}")


if __name__ == "__main__":
    main()
