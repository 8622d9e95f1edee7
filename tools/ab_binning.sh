# A-B of the one-pass binning (temporary helper):  bash tools/ab_binning.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_raster_gpu.py tests/test_speculation_gpu.py tests/test_raster_parity_gpu.py -q -x 2>&1 | tail -5 > $O/pytest.txt
RTGS_BIN_STAGE=0 timeout 600 python -m pytest tests/test_raster_gpu.py -q -x -k "onepass or near_slice" 2>&1 | tail -3 > $O/pytest_direct.txt
for v in onepass old; do
  export RTGS_BIN_ONEPASS=1
  [ $v = old ] && export RTGS_BIN_ONEPASS=0
  for w in headline surface; do
    python $R/tools/prof_raster.py $w 20 > $O/plain_${v}_$w.txt 2>&1
  done
done
export RTGS_BIN_ONEPASS=1
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cat $O/pytest.txt $O/pytest_direct.txt; tail -qn 1 $O/plain_*.txt; python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['repeats'],d['frames_per_sec_replica_schedule'],d['surface_scene']['map_iteration_ms'],d['strong_scaling_one_view']['ms_per_iteration'], d['roofline']['kernel'], d['roofline']['frac'])"
