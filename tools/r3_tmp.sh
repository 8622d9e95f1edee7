cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_icp_gpu.py tests/test_icp_stream_gpu.py tests/test_slam_stream_gpu.py -x -q -k "not full_size_vs_reference" 2>&1 | tail -4
python tools/prof_icp.py replica 50 2>&1 | tail -1
python tools/prof_icp.py tum 50 2>&1 | tail -1
