# One gpurun call that produces everything profiles/ quotes for a round:
#   bash tools/measure_round.sh r04            -> gpurun_out/r04/*
# GPU tests, the default bench line, kernel traces (bench, map iteration of both scenes, ICP), gap traces, and the PMC
# passes - FETCH_SIZE and WRITE_SIZE in separate runs (TCC slots), two SQ counter sets - each with --kernel-trace only.
set -x
R=$GRAFT_REPO_ROOT
TAG=${1:-r05}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu.txt
fi
python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_bench -o bench -- python $R/bench.py --no-cpu-baseline --no-schedule --no-surface --prewarm 400 > $O/ks_bench.log 2>&1
python $R/tools/unit_trace.py $(find $O/ks_bench -name "*kernel_trace.csv" | head -1) 150 200 > $O/unit_trace.txt 2>&1
for w in headline surface; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$w -o k -- python $R/tools/prof_raster.py $w 50 > $O/ks_$w.log 2>&1
  python $R/tools/kernel_table.py $O/ks_$w 30 > $O/table_$w.txt
  python $R/tools/gap_trace.py $(find $O/ks_$w -name "*kernel_trace.csv" | head -1) map_fused_tail 25 > $O/gaps_$w.txt
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_$w -o f -- python $R/tools/prof_raster.py $w 5 > $O/pmc_f_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_$w -o w -- python $R/tools/prof_raster.py $w 5 > $O/pmc_w_$w.log 2>&1
  python $R/tools/traffic_from_pmc.py $(find $O/pmc_f_$w -name "*counter_collection.csv" | head -1) $(find $O/pmc_w_$w -name "*counter_collection.csv" | head -1) $O/traffic_$w.json > $O/traffic_$w.txt
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_icp -o k -- python $R/tools/prof_icp.py replica 40 > $O/ks_icp.log 2>&1
# round 5: the SLAM sequence (BASELINE configs[2]), the unchanged-reference iteration and BASELINE configs[4] under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_seq -o k -- python $R/bench.py --only sequence --sequence-frames 300 > $O/ks_seq.log 2>&1
python $R/tools/kernel_table.py $O/ks_seq 45 > $O/table_sequence.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_dropin -o k -- python $R/bench.py --only dropin --steps 10 > $O/ks_dropin.log 2>&1
python $R/tools/kernel_table.py $O/ks_dropin 45 > $O/table_dropin.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c5 -o k -- python $R/bench.py --only config5 > $O/ks_c5.log 2>&1
python $R/tools/kernel_table.py $O/ks_c5 30 > $O/table_config5.txt
RTGS_MAP_PROFILE=1 python $R/bench.py --only sequence --sequence-frames 300 > $O/sequence_stage_profile.json 2> /dev/null
$R/tools/probe/valu_rate > $O/valu_rate.txt 2>&1
bash $R/tools/pmc_sq_passes.sh $TAG/sq > $O/pmc_sq.log 2>&1
python $R/tools/valu_from_pmc.py $O/sq/pmc_sq_surface.csv $O/sq/pmc_sq_headline.csv $O/valu.json
for w in headline surface; do python $R/tools/pmc_summary.py $O/sq/pmc_sq_$w.csv > $O/sq_summary_$w.txt; done
# keep the merge small: raw traces out, the --stats tables stay
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
cat $O/pytest_gpu.txt; tail -c 800 $O/bench.json; cat $O/gaps_headline.txt | head -3; cat $O/gaps_surface.txt | head -3; du -sh $O
