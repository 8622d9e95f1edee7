set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05b
mkdir -p $O
cd $R
timeout 100 tools/probe/valu_rate > $O/valu_rate.txt 2> $O/valu_rate.err
timeout 300 python -m pytest tests/test_trainable_gpu.py -m gpu -q -k "global_optimization" 2>&1 | tail -30 > $O/t_global.txt
timeout 400 python -m pytest tests/test_dist_gpu.py -m gpu -q -k "tile_band" 2>&1 | tail -30 > $O/t_band.txt
timeout 600 python -m pytest tests/test_bench_launcher_gpu.py -m gpu -q -x 2>&1 | tail -30 > $O/t_launch.txt
RTGS_MAP_PROFILE=1 timeout 300 python bench.py --only sequence --sequence-frames 150 > $O/seq150_prof.json 2> $O/seq150_prof.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_seq -o k -- python $R/bench.py --only sequence --sequence-frames 150 > $O/ks_seq.log 2>&1
cd $R
python tools/kernel_table.py $O/ks_seq 40 > $O/table_seq.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -3 $O/t_global.txt; tail -8 $O/t_band.txt; tail -12 $O/t_launch.txt; tail -c 1200 $O/seq150_prof.json; head -45 $O/table_seq.txt; tail -3 $O/ks_seq.log; tail -c 3000 $O/bench.json; tail -5 $O/bench.err; tail -25 $O/valu_rate.txt; cat $O/valu_rate.err
