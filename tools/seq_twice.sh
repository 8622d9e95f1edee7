#!/bin/bash
# the 2 000-frame sequence N times (default 2) + the lifecycle / map tests:  tools/seq_twice.sh <label> [ENV=..]
L=${1:-x}; shift
R=$(pwd); O=$R/gpurun_out/r06_sq_$L; mkdir -p $O
python -m pytest tests/test_slam_ops_gpu.py tests/test_sequence_gpu.py tests/test_trainable_gpu.py tests/test_slam_stream_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
cd /tmp
for rep in 1 2 3; do
  env "$@" python $R/bench.py --only sequence --sequence-frames 2000 > $O/seq.$rep.json 2> $O/seq.$rep.err
  python - <<PY
import json
s=json.load(open("$O/seq.$rep.json"))["sequence"]
print("rep=$rep", {k:s[k] for k in ("fps","fps_tracking_plus_mapping","mapping_ms_mean_optimised_frames","mapping_ms_mean_other_frames","tracking_ms_mean","gaussians","ate_rmse_m")}, s["stats"]["added"])
PY
done
