set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05x
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ord in 0 1; do for w in headline surface; do
  RTGS_BWD_ORDER=$ord timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_${w}_$ord -o k -- python $R/tools/prof_raster.py $w 20 > $O/log_${w}_$ord.txt 2>&1
  python $R/tools/kernel_table.py $O/ks_${w}_$ord 12 > $O/table_${w}_$ord.txt
done; done
cd $R
RTGS_BWD_ORDER=1 timeout 300 python -m pytest tests/test_bwd_walks_gpu.py -m gpu -q 2>&1 | tail -3 > $O/t_walks.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
for f in $O/table_*; do echo $f; grep -E "blend_bwd_mfma|tile_order|blend_fwd|map_fused" $f; done; grep -h "iter " $O/log_*; cat $O/t_walks.txt
