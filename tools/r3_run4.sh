set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3d}; mkdir -p $O
timeout 1500 python -m pytest tests/test_bwd_walks_gpu.py tests/test_raster_gpu.py tests/test_raster_parity_gpu.py -x -q 2>&1 | tail -15 > $O/pytest.txt
cat $O/pytest.txt
python tools/walk_stats.py both > $O/walk_stats.txt 2>&1; cat $O/walk_stats.txt; for w in headline surface; do python tools/prof_raster.py $w 30 2>&1 | tail -1 >> $O/prof.txt; done
cat $O/prof.txt
