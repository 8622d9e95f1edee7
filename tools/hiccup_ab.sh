#!/bin/bash
# How often does a unit's host frame hiccup?  tools/hiccup_ab.sh  (through gpurun): 150 blocks of 20 units per GIL switch interval
R=$(pwd); O=$R/gpurun_out/r06_hiccup; mkdir -p $O
for us in 5000 50; do   # (0 would leave the interpreter default)
  RTGS_GIL_SWITCH_US=$us python bench.py --repeats 150 --no-cpu-baseline --no-surface --no-sequence --no-config5 --no-dropin --no-schedule > $O/b_$us.json 2> $O/b_$us.err
  python - <<PY
import json
d=json.load(open("$O/b_$us.json"))
r=d["repeats"]; h=r["slowest_host_frame_ms_per_block"]; b=r["ms_per_step"]
print(r["host_frames_over_2ms"]); print("switch interval $us us: blocks", len(b), "median %.4f" % r["median_ms_per_step"], "blocks > 1.05 x median:", sum(x > 1.05*r["median_ms_per_step"] for x in b), "host frames > 2 ms:", sum(x > 2 for x in h), "max host frame %.2f" % max(h), "first block %.4f" % b[0])
PY
done
