#!/bin/bash
# kernel-by-kernel timeline of ONE local-optimisation iteration late in a 120-frame sequence (start, duration, gap before)
R=$(pwd); O=$R/gpurun_out/r06_itertrace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o k -- python $R/bench.py --only sequence --sequence-frames 120 > $O/run.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
tails = [i for i, n in enumerate(names) if n.startswith("map_fused_tail") or "map_fused_tail" in n]
# the 30th-from-last pair of consecutive tails = one iteration in the middle of a late local optimisation
out = open("$O/iteration.txt", "w")
for pick in (len(tails) - 30, len(tails) - 12):
    a, b = tails[pick], tails[pick + 1]
    t0 = int(rows[a]["End_Timestamp"]); prev = t0
    out.write("---- iteration between tail #%d and #%d: %d kernels, %.1f us\n" % (pick, pick + 1, b - a, (int(rows[b]["End_Timestamp"]) - t0) / 1e3))
    for r in rows[a + 1:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        out.write("%8.1f  gap %6.1f  dur %6.1f  %s\n" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:70]))
        prev = e
out.close()
print(open("$O/iteration.txt").read())
PY
find $O/kt -name "*kernel_trace.csv" -delete
