#!/bin/bash
# One measurement step of round 6 (run through gpurun):  tools/r06_step.sh <label> [tests]
#   kernel tables of the map iteration on both 1.2 M scenes, the per-wave stamps of both blend kernels, optional quick tests.
L=${1:-x}; R=$(pwd); O=$R/gpurun_out/r06_$L; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ -n "$2" ]; then
  (cd $R && timeout 900 python -m pytest $2 -x -q -m gpu 2>&1 | tail -15) > $O/tests.txt 2>&1
  tail -5 $O/tests.txt
fi
for w in headline surface; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$w -o k -- python $R/tools/prof_raster.py $w 30 > $O/prof_$w.txt 2>&1
  python $R/tools/kernel_table.py $O/ks_$w 16 > $O/table_$w.txt 2>&1
  head -16 $O/table_$w.txt; tail -3 $O/prof_$w.txt
  find $O/ks_$w -name "*kernel_trace.csv" -delete
done
for w in headline surface; do
  timeout 250 python $R/tools/bwd_stamps.py $w > $O/bwd_stamps_$w.txt 2>&1; grep -E "span|share|prologue|lifetime|list sched" $O/bwd_stamps_$w.txt
  timeout 250 python $R/tools/fwd_stamps.py $w > $O/fwd_stamps_$w.txt 2>&1; grep -E "span|share|lifetime|per wave|list sched" $O/fwd_stamps_$w.txt
done
