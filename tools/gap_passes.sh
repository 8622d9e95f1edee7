set -x
R=$GRAFT_REPO_ROOT; TAG=${1:-r3gaps}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in headline surface; do
  rocprofv3 --kernel-trace --output-format csv -d $O/kt_$w -o k -- python $R/tools/prof_raster.py $w 40 > $O/kt_$w.log 2>&1
  python $R/tools/gap_trace.py $(find $O/kt_$w -name "*kernel_trace.csv" | head -1) map_tail_rows 25 > $O/gaps_$w.txt
  cat $O/gaps_$w.txt
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
