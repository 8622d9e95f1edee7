set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05e
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_slam_ops_gpu.py tests/test_sequence_gpu.py -m gpu -q 2>&1 | tail -30 > $O/t1.txt
timeout 300 python bench.py --only sequence --sequence-frames 400 > $O/seq400.json 2> $O/seq400.err
RTGS_MAP_PROFILE=1 timeout 300 python bench.py --only sequence --sequence-frames 150 > $O/seq150_prof.json 2> $O/seq150_prof.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_seq -o k -- python $R/bench.py --only sequence --sequence-frames 150 > $O/ks_seq.log 2>&1
cd $R
python tools/kernel_table.py $O/ks_seq 200 > $O/table_seq.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
tail -6 $O/t1.txt; python -c "
import json
d=json.load(open('$O/seq150_prof.json'))['sequence']; print(d['fps'], d['stage_profile_ms_per_frame'])
d=json.load(open('$O/seq400.json'))['sequence']; print({k:d[k] for k in ('fps','fps_tracking_plus_mapping','ate_rmse_m','gaussians','mapping_ms_mean_optimised_frames','mapping_ms_mean_other_frames','tracking_ms_mean')})
"
head -14 $O/table_seq.txt; python - <<'PY'
import csv,sys
rows=[l.split() for l in open('/root/repo/gpurun_out/r05e/table_seq.txt').read().splitlines()[1:] if l.strip()]
calls=sum(int(r[-3]) for r in rows); tot=sum(float(r[-1]) for r in rows)
print("kernel launches", calls, "total kernel ms", round(tot,1), "per frame", calls/150, tot/150)
PY
