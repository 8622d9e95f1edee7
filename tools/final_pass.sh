# Final pass of the round on the final kernel sources: PMC re-stamp (FETCH / WRITE / SQ, headline + surface), smoke, full
# GPU suite, then the bench line with the fresh stamps.   bash tools/final_pass.sh
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in headline surface; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_$w -o f -- python $R/tools/prof_raster.py $w 5 > $O/pmc_f_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_$w -o w -- python $R/tools/prof_raster.py $w 5 > $O/pmc_w_$w.log 2>&1
  python $R/tools/traffic_from_pmc.py $(find $O/pmc_f_$w -name "*counter_collection.csv" | head -1) $(find $O/pmc_w_$w -name "*counter_collection.csv" | head -1) $O/traffic_$w.json > $O/traffic_$w.txt
done
bash $R/tools/pmc_sq_passes.sh r03f/sq > $O/pmc_sq.log 2>&1
python $R/tools/valu_from_pmc.py $O/sq/pmc_sq_surface.csv $O/sq/pmc_sq_headline.csv $O/valu.json
cp $O/traffic_headline.json $R/profiles/traffic_latest.json; cp $O/valu.json $R/profiles/valu_latest.json
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
du -sh $O
