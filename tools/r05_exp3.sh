set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dbg in 1 5; do for w in headline surface; do
  RTGS_MFMA_DEBUG=$dbg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_${w}_$dbg -o k -- python $R/tools/prof_raster.py $w 10 > $O/log_${w}_$dbg.txt 2>&1
  python $R/tools/kernel_table.py $O/ks_${w}_$dbg 12 > $O/table_${w}_$dbg.txt
done; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
for f in $O/table_*; do echo $f; grep -E "blend_bwd_mfma" $f; done
