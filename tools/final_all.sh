# Everything the committed profiles quote, on the final tree, in one gpurun call:  bash tools/final_all.sh
#   smoke, GPU suite, kernel tables (headline / surface), PMC passes + bench line (tools/pmc_final.sh)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
for w in headline surface; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$w -o k -- python $R/tools/prof_raster.py $w 30 > $O/prof_$w.log 2>&1
  python $R/tools/kernel_table.py $O/ks_$w 40 > $O/kernel_table_$w.txt
  cp $(find $O/ks_$w -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$w.csv
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cd $R
bash tools/pmc_final.sh
