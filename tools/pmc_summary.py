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
        # What these counters can and cannot say: SQ_ACTIVE_INST_VALU ticks ONCE per wave-instruction in
        # quad-cycle units - SQ_ACTIVE_INST_VALU x 4 / SQ_INSTS_VALU is 4.00 for every kernel, a definition, not a
        # measurement - so "VALU-active cycles" derived from it only restate the instruction count.  What IS measured:
        # instructions per SIMD, busy cycles, and hence instructions per busy SIMD-cycle, to hold against the issue rates
        # of tools/probe/valu_rate.hip (profiles/r05_valu_rate.txt: one wave64 VALU per ~1.4 cycles at 8 waves / SIMD for
        # plain FMA, ~2.4 for DPP, ~4.6 for v_exp_f32).  SQ_THREAD_CYCLES_VALU counts lit lanes per instruction in the same
        # units: lanes lit = THREAD_CYCLES / (ACTIVE_INST x 64) (round 4 divided by another 4 and topped out at 0.25).
        per_simd = c["SQ_INSTS_VALU"] / 1024
        busy = c["SQ_BUSY_CYCLES"] / 32
        print(f"{k:22s} VALU {c['SQ_INSTS_VALU'] / 1e6:7.2f}M SALU {c['SQ_INSTS_SALU'] / 1e6:6.2f}M LDS {c['SQ_INSTS_LDS'] / 1e6:6.2f}M | "
              f"busy {busy / 1e3:6.0f}k cyc, {per_simd / 1e3:6.1f}k VALU/SIMD = {per_simd / busy:.3f} VALU per busy SIMD-cycle "
              f"({busy / max(per_simd, 1):.1f} cyc/VALU; probe floor 1.4), lanes lit {c['SQ_THREAD_CYCLES_VALU'] / (c['SQ_ACTIVE_INST_VALU'] * 64):.2f}, "
              f"wait_inst {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.2f} wait_any {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.2f} "
              f"waves {c['SQ_WAVES']:.0f}")


if __name__ == "__main__":
    main()
