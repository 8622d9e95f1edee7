set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05c
mkdir -p $O
cd $R
timeout 100 tools/probe/valu_rate > $O/valu_rate.txt 2> $O/valu_rate.err
timeout 300 python -m pytest tests/test_trainable_gpu.py tests/test_slam_ops_gpu.py -m gpu -q -k "global_optimization or knn" 2>&1 | tail -30 > $O/t_global.txt
timeout 400 python -m pytest tests/test_dist_gpu.py -m gpu -q -k "tile_band" 2>&1 | tail -30 > $O/t_band.txt
RTGS_MAP_PROFILE=1 timeout 300 python bench.py --only sequence --sequence-frames 150 > $O/seq150_prof.json 2> $O/seq150_prof.err
timeout 300 python bench.py --only sequence --sequence-frames 400 > $O/seq400.json 2> $O/seq400.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_seq -o k -- python $R/bench.py --only sequence --sequence-frames 150 > $O/ks_seq.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_dropin -o k -- python $R/bench.py --only dropin --steps 10 > $O/ks_dropin.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c5 -o k -- python $R/bench.py --only config5 > $O/ks_c5.log 2>&1
cd $R
for w in seq dropin c5; do python tools/kernel_table.py $O/ks_$w 45 > $O/table_$w.txt 2>&1; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
tail -3 $O/t_global.txt; tail -8 $O/t_band.txt; python -c "
import json
d=json.load(open('$O/seq150_prof.json'))['sequence']; print(d['fps'], d['stage_profile_ms_per_frame'])
d=json.load(open('$O/seq400.json'))['sequence']; print({k:d[k] for k in ('fps','fps_tracking_plus_mapping','ate_rmse_m','gaussians','mapping_ms_mean_optimised_frames','mapping_ms_mean_other_frames','tracking_ms_mean')})
"
head -30 $O/table_seq.txt; grep -h "dropin\|config5" $O/ks_dropin.log $O/ks_c5.log | tail -3; head -40 $O/table_dropin.txt; head -25 $O/table_c5.txt; tail -9 $O/valu_rate.txt; cat $O/valu_rate.err
