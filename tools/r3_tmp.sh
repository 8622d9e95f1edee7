cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-surface --no-schedule 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prio -1:', d['value'], d['repeats']['ms_per_step'], d['icp_track_ms'], d['strong_scaling_one_view']['ms_per_iteration'])"
sed -i 's/tracker_priority: int = -1/tracker_priority: int = 0/' rtg_slam_amd/pipeline.py
python bench.py --no-cpu-baseline --no-surface --no-schedule 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prio 0:', d['value'], d['repeats']['ms_per_step'], d['icp_track_ms'], d['strong_scaling_one_view']['ms_per_iteration'])"
