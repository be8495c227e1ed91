cd $GRAFT_REPO_ROOT
python tools/probes/_dbg.py 2>&1 | tail -4
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
