cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q --durations=25 2>&1 | tail -40
