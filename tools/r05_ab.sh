#!/bin/bash
# A-B timing of the backward walk: RTGS_MFMA_DEBUG=8 (MFMA sums, round 4) against 0 (lane sums), then the stamps of the default
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05p; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for dbg in ${DBGS:-8 0}; do for w in headline surface; do
  RTGS_MFMA_DEBUG=$dbg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_${w}_$dbg -o k -- python $R/tools/prof_raster.py $w 10 > /dev/null 2>&1
  python $R/tools/kernel_table.py $O/ks_${w}_$dbg 14 > $O/table_${w}_$dbg.txt
done; done
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
for f in $O/table_*; do echo $f; grep -E "blend_bwd" $f; done
for w in headline surface; do timeout 250 python $R/tools/mfma_stamps.py $w > $O/stamps_$w.txt 2>&1; grep -E "span|cycles|lifetime" $O/stamps_$w.txt; done
