#!/bin/bash
# timing decomposition of the backward walk: kernel time with parts switched off (rtgs_raster_set_bwd_debug bits)
L=${1:-x}; R=$(pwd); O=$R/gpurun_out/r06_$L; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in headline surface; do
for dbg in 0 1 2 4 3 7; do
  RTGS_BWD_DEBUG_TOOL=$dbg timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dc_${w}_$dbg -o k -- python $R/tools/prof_raster.py $w 10 > /dev/null 2>&1
  echo "$w dbg=$dbg: $(python $R/tools/kernel_table.py $O/dc_${w}_$dbg 40 | grep -E 'blend_bwd_entry|bwd_order' | awk '{print $1, $3}' | tr '\n' ' ')"
  find $O/dc_${w}_$dbg -name "*kernel_trace.csv" -delete
done; done 2>&1 | tee $O/decomp.txt
for w in headline surface; do
  timeout 250 python $R/tools/bwd_stamps.py $w > $O/bwd_stamps_$w.txt 2>&1; grep -E "span|share|prologue|lifetime|list sched" $O/bwd_stamps_$w.txt
done
