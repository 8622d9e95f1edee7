#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (ROCm 7.2 default output) into the same table
`--stats` prints: per-kernel calls / total / average / min / max duration.
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.csv]
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["name,calls,total_us,avg_us,min_us,max_us,pct"]
    for n, k, s, a, mn, mx in rows:
        n = n.replace(",", ";")
        lines.append(f"\"{n[:110]}\",{k},{s / 1e3:.1f},{a / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100 * s / tot:.2f}")
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
