set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05f
mkdir -p $O
cd $R
timeout 600 python bench.py --no-cpu-baseline --no-surface --no-schedule --no-sequence --no-dropin --prewarm 200 --steps 5 --repeats 1 > $O/bench_c5.json 2> $O/bench_c5.err
timeout 300 python bench.py --only config5 > $O/only_c5.json 2> $O/only_c5.err
timeout 900 python -m pytest tests/test_raster_parity_gpu.py -m gpu -q -x -s 2>&1 | tail -25 > $O/t_parity.txt
python - <<'PY'
import json
for f in ("bench_c5","only_c5"):
    try:
        d=json.load(open(f"/root/repo/gpurun_out/r05f/{f}.json")); d=d.get("config5", d)
        print(f, json.dumps(d)[:1200])
    except Exception as e: print(f, "ERR", e)
PY
tail -5 $O/bench_c5.err; tail -25 $O/t_parity.txt
