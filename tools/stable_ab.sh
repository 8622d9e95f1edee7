#!/bin/bash
# A/B of the remembered neighbour-search structure over the stable Gaussians (RTGS_STABLE_SEARCH) + the tests around it
R=$(pwd); O=$R/gpurun_out/r06_stable; mkdir -p $O
python -m pytest tests/test_slam_ops_gpu.py tests/test_sequence_gpu.py tests/test_slam_stream_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
cd /tmp
for rep in 1 2; do
for v in 0 1; do
  RTGS_STABLE_SEARCH=$v python $R/bench.py --only sequence --sequence-frames 2000 > $O/seq_$v.$rep.json 2> $O/seq_$v.$rep.err
  python - <<PY
import json
s=json.load(open("$O/seq_$v.$rep.json"))["sequence"]
print("stable_search=$v rep=$rep", {k:s[k] for k in ("fps","mapping_ms_mean_optimised_frames","mapping_ms_mean_other_frames","gaussians","ate_rmse_m")}, s["stats"]["added"])
PY
done; done
