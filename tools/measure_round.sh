set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_bench -o bench -- python $R/bench.py --no-cpu-baseline > $O/ks_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_head -o head -- python $R/tools/prof_raster.py headline 50 > $O/ks_head.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_surf -o surf -- python $R/tools/prof_raster.py surface 50 > $O/ks_surf.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_head -o f -- python $R/tools/prof_raster.py headline 5 > $O/pmc_f_head.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_head -o w -- python $R/tools/prof_raster.py headline 5 > $O/pmc_w_head.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_surf -o f -- python $R/tools/prof_raster.py surface 5 > $O/pmc_f_surf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_surf -o w -- python $R/tools/prof_raster.py surface 5 > $O/pmc_w_surf.log 2>&1
cd $R
find $O -name "*counter_collection.csv" | head
python tools/traffic_from_pmc.py $(find $O/pmc_f_head -name "*counter_collection.csv") $(find $O/pmc_w_head -name "*counter_collection.csv") $O/traffic_head.json
python tools/traffic_from_pmc.py $(find $O/pmc_f_surf -name "*counter_collection.csv") $(find $O/pmc_w_surf -name "*counter_collection.csv") $O/traffic_surf.json
# keep the merge small: drop raw traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
cat $O/pytest_gpu.txt; tail -c 600 $O/bench.json; du -sh $O
