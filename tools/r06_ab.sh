#!/bin/bash
# A-B of kernel variants selected by environment variables:  tools/r06_ab.sh <label> "<VAR=a VAR=b ...>" [tests]
#   (4th argument: the settings to take backward stamps with)
#   for every setting: kernel table of the map iteration on both 1.2 M scenes (blend kernels + order kernel), then backward stamps
L=${1:-x}; R=$(pwd); O=$R/gpurun_out/r06_$L; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ -n "$3" ]; then
  (cd $R && timeout 900 python -m pytest $3 -x -q -m gpu 2>&1 | tail -15) > $O/tests.txt 2>&1
  tail -3 $O/tests.txt
fi
for setting in $2; do
  for w in headline surface; do
    env ${setting//,/ } timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_${w}_$setting -o k -- python $R/tools/prof_raster.py $w 20 > $O/prof_${w}_$setting.txt 2>&1
    echo "$setting $w: $(python $R/tools/kernel_table.py $O/ks_${w}_$setting 40 | grep -E 'blend_|bwd_order|map_fused' | awk '{print $1, $3}' | tr '\n' ' ') | $(grep 'un-profiled' $O/prof_${w}_$setting.txt | cut -d: -f2 | cut -d'|' -f1)"
    find $O/ks_${w}_$setting -name "*kernel_trace.csv" -delete
  done
done 2>&1 | tee $O/ab.txt
for setting in $4; do
  for w in headline surface; do
    env ${setting//,/ } timeout 250 python $R/tools/bwd_stamps.py $w > $O/bwd_stamps_${w}_$setting.txt 2>&1; echo "== $setting"; grep -E "span|share|prologue|lifetime|list sched" $O/bwd_stamps_${w}_$setting.txt
  done
done
