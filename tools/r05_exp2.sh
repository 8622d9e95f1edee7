set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05y
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_trainable_gpu.py tests/test_raster_gpu.py tests/test_sequence_gpu.py -m gpu -q -x 2>&1 | tail -8 > $O/t1.txt
cd /tmp && export TMPDIR=/tmp
for w in headline surface; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$w -o k -- python $R/tools/prof_raster.py $w 20 > $O/log_$w.txt 2>&1
  python $R/tools/kernel_table.py $O/ks_$w 12 > $O/table_$w.txt
done
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cat $O/t1.txt; for f in $O/table_*; do echo $f; grep -E "blend_bwd_mfma|blend_fwd|map_fused" $f; done; grep -h "iter \|blocks\|ms per" $O/log_* | tail -8
