# PMC passes again on the final kernel sources (bench.py quotes roofline.traffic / valu only from tables stamped with the
# current source hash), then the committed bench line:  bash tools/pmc_final.sh  -> gpurun_out/pmcf/*
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcf
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in headline surface; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_$w -o f -- python $R/tools/prof_raster.py $w 5 > $O/pmc_f_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_$w -o w -- python $R/tools/prof_raster.py $w 5 > $O/pmc_w_$w.log 2>&1
  python $R/tools/traffic_from_pmc.py $(find $O/pmc_f_$w -name "*counter_collection.csv" | head -1) $(find $O/pmc_w_$w -name "*counter_collection.csv" | head -1) $O/traffic_$w.json > $O/traffic_$w.txt
done
bash $R/tools/pmc_sq_passes.sh pmcf/sq > $O/pmc_sq.log 2>&1
python $R/tools/valu_from_pmc.py $O/sq/pmc_sq_surface.csv $O/sq/pmc_sq_headline.csv $O/valu.json
for w in headline surface; do python $R/tools/pmc_summary.py $O/sq/pmc_sq_$w.csv > $O/sq_summary_$w.txt; done
cp $O/traffic_headline.json $R/profiles/traffic_latest.json; cp $O/valu.json $R/profiles/valu_latest.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/pmcf/bench.json'))
print({k:d[k] for k in ("metric","value","unit","ms_per_step","slam_frames_per_sec","dropin_iteration_ms","unstable")})
print(d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["valu"], d["roofline"]["bound"])
PY
