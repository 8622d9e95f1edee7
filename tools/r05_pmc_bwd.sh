#!/bin/bash
# what a launch of the backward walk keeps busy: instruction fetch, LDS, scalar unit, VALU / MFMA cycles (separate PMC passes)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05q; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  for dbg in ${DBGS:-8 0}; do
    RTGS_MFMA_DEBUG=$dbg timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p${i}_$dbg -o p -- python $R/tools/prof_raster.py headline 3 > $O/p${i}_$dbg.log 2>&1
  done
done
for dbg in ${DBGS:-8 0}; do
  args=""; for j in 1 2 3 4 5 6; do f=$(find $O/p${j}_$dbg -name "*counter_collection.csv" | head -1); [ -n "$f" ] && args="$args s$j=$f"; done
  python $R/tools/pmc_sq.py $O/pmc_$dbg.csv $args
  grep "blend_bwd" $O/pmc_$dbg.csv
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
