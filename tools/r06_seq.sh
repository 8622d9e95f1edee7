#!/bin/bash
# The SLAM sequence (BASELINE configs[2]) under the kernel trace + its stage profile:  tools/r06_seq.sh <label> [frames] [ENV=.. ENV=..]
L=${1:-x}; F=${2:-300}; shift; shift
R=$(pwd); O=$R/gpurun_out/r06_seq_$L; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env "$@" python $R/bench.py --only sequence --sequence-frames $F > $O/seq.json 2> $O/seq.err
env "$@" RTGS_MAP_PROFILE=1 python $R/bench.py --only sequence --sequence-frames $F > $O/stage_profile.json 2> /dev/null
env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o k -- python $R/bench.py --only sequence --sequence-frames $F > $O/ks.log 2>&1
python $R/tools/kernel_table.py $O/ks 60 > $O/table_sequence.txt
find $O/ks -name "*kernel_trace.csv" -delete
python - <<PY
import json
s=json.load(open("$O/seq.json"))["sequence"]
print({k:s[k] for k in ("frames","fps","fps_tracking_plus_mapping","mapping_ms_mean_optimised_frames","mapping_ms_mean_other_frames","tracking_ms_mean","peak_device_memory_MB","gaussians","ate_rmse_m")})
p=json.load(open("$O/stage_profile.json"))["sequence"]["stage_profile_ms_per_frame"]; print(p)
PY
head -45 $O/table_sequence.txt; tail -2 $O/table_sequence.txt
