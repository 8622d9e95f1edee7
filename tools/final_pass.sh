# The committed bench line + smoke on the final tree:  bash tools/final_pass.sh  -> gpurun_out/final/{bench.json,smoke.txt}
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
if [ "${SKIP_TESTS:-0}" != "1" ]; then timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt; fi
python bench.py > $O/bench.json 2> $O/bench.err
tail -2 $O/smoke.txt; python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/final/bench.json'))
print({k:d[k] for k in ("metric","value","unit","ms_per_step","slam_frames_per_sec","dropin_iteration_ms","frames_per_sec_replica_schedule","unstable")})
print(d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["valu"], d["roofline"]["bound"])
print(d["config5"]["ms_per_iteration"], d["config5"]["split_ms"], d["config5"]["sparse_form_ms_per_iteration"])
s=d["slam_sequence"]; print({k:s[k] for k in ("fps","fps_tracking_plus_mapping","ate_rmse_m","gaussians","stable","mapping_ms_mean_optimised_frames","mapping_ms_mean_other_frames","tracking_ms_mean","peak_device_memory_MB")})
PY
