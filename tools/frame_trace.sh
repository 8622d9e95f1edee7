#!/bin/bash
# GPU timeline of the last frames of a sequence: the tail of the kernel trace comes back as a small CSV (start, end, name)
R=$(pwd); O=$R/gpurun_out/r06_frametrace; mkdir -p $O; F=${1:-700}
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o k -- python $R/bench.py --only sequence --sequence-frames $F > $O/run.log 2>&1
python - <<PY 2>&1
import csv, glob
f = glob.glob("$O/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(len(rows), "kernels")
with open("$O/tail.csv", "w") as o:
    for r in rows[-12000:]:
        o.write("%s,%s,%s\n" % (r["Start_Timestamp"], r["End_Timestamp"], r["Kernel_Name"].replace(",", ";")[:80]))
PY
find $O/kt -name "*kernel_trace.csv" -delete
