set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
timeout 120 tools/probe/valu_rate > $O/valu_rate.txt 2>&1
timeout 300 python -m pytest tests/test_slam_ops_gpu.py -m gpu -q -k "knn" -x 2>&1 | tail -15 > $O/t_knn.txt
timeout 400 python -m pytest tests/test_trainable_gpu.py -m gpu -q -k "frozen_map or global_optimization" 2>&1 | tail -40 > $O/t_global.txt
timeout 500 python -m pytest tests/test_sequence_gpu.py -m gpu -q -s 2>&1 | tail -60 > $O/t_seq.txt
timeout 300 python bench.py --only sequence --sequence-frames 150 > $O/seq150.json 2> $O/seq150.err
timeout 120 python bench.py --only icp_tum > $O/icp_tum.json 2> $O/icp_tum.err
tail -5 $O/t_knn.txt; tail -25 $O/t_global.txt; tail -30 $O/t_seq.txt; tail -c 1500 $O/seq150.json; tail -5 $O/seq150.err; cat $O/icp_tum.json; head -12 $O/valu_rate.txt
