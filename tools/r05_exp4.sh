set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05w
mkdir -p $O
cd $R
timeout 300 python bench.py --only sequence --sequence-frames 100 > /dev/null 2>&1
for rep in 1 2; do for lm in 100000 20000; do
  RTGS_LARGE_MAP_MIN=$lm timeout 300 python bench.py --only sequence --sequence-frames 400 > $O/seq_${lm}_$rep.json 2> $O/seq_$lm.err
done; done
cd /tmp && export TMPDIR=/tmp
RTGS_LARGE_MAP_MIN=20000 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_seq -o k -- python $R/bench.py --only sequence --sequence-frames 150 > $O/ks_seq.log 2>&1
cd $R
python tools/kernel_table.py $O/ks_seq 16 > $O/table_seq.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
python -c "
import json
for rep in (1,2):
  for lm in (100000,20000):
    d=json.load(open('$O/seq_%d_%d.json'%(lm,rep)))['sequence']; print(lm, rep, {k:d[k] for k in ('fps','fps_tracking_plus_mapping','ate_rmse_m','gaussians','mapping_ms_mean_optimised_frames','mapping_ms_mean_other_frames')}, d['speculation'])
"
cat $O/table_seq.txt
