# SQ counter passes over the map iteration of both 1.2 M scenes (own runs, --pmc only + kernel trace):
#   bash tools/pmc_sq_passes.sh <tag>      -> gpurun_out/<tag>/pmc_sq_{headline,surface}.csv
set -x
R=$GRAFT_REPO_ROOT; TAG=${1:-r3pmc}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
S1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
S2="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM"
for w in headline surface; do
  rocprofv3 --pmc $S1 --kernel-trace --output-format csv -d $O/s1_$w -o s -- python $R/tools/prof_raster.py $w 5 > $O/s1_$w.log 2>&1
  rocprofv3 --pmc $S2 --kernel-trace --output-format csv -d $O/s2_$w -o s -- python $R/tools/prof_raster.py $w 5 > $O/s2_$w.log 2>&1
  python $R/tools/pmc_sq.py $O/pmc_sq_$w.csv set1=$(find $O/s1_$w -name "*counter_collection.csv" | head -1) set2=$(find $O/s2_$w -name "*counter_collection.csv" | head -1)
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
tail -3 $O/s1_surface.log $O/s2_surface.log; grep -c . $O/pmc_sq_*.csv; du -sh $O
