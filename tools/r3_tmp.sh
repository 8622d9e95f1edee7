cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_raster_parity_gpu.py -q -k "config5" 2>&1 | tail -3
timeout 600 python bench.py --gaussians 5000000 --mode sharded --no-surface --no-schedule --no-cpu-baseline --steps 10 --prewarm 20 2>&1 | tail -c 1500
