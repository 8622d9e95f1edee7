set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3g}; mkdir -p $O
timeout 1500 python -m pytest tests/test_speculation_gpu.py tests/test_raster_gpu.py tests/test_context_gpu.py tests/test_bwd_walks_gpu.py -x -q 2>&1 | tail -25 > $O/pytest.txt
cat $O/pytest.txt
for sp in 1 0; do for w in headline surface; do RTGS_SPECULATE=$sp python tools/prof_raster.py $w 30 2>&1 | tail -1 | sed "s|^|spec=$sp |" >> $O/prof.txt; done; done
cat $O/prof.txt
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
