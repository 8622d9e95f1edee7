cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_raster_gpu.py tests/test_speculation_gpu.py -x -q -k "normal or one_call or speculat" 2>&1 | tail -12
