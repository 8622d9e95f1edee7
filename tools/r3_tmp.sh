cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
python bench.py > gpurun_out/r03/bench_final.json 2> gpurun_out/r03/bench_final.err; tail -c 600 gpurun_out/r03/bench_final.json
