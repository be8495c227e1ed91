#!/bin/bash
# grid tunables after the round-5 regrouping: yz cells per coarse cell edge / x refinement / queries per coarse cell
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for spec in "2.0 2.0 2.0" "2.5 2.0 2.0" "3.0 2.0 2.0" "1.5 2.0 2.0" "2.0 1.0 2.0" "2.0 3.0 2.0" "2.0 4.0 2.0" "2.5 1.5 2.0" "2.0 2.0 1.0" "2.0 2.0 4.0" "2.5 2.0 1.0"; do
  set -- $spec
  for c in 2 3; do
    DEFTET_PIT_YZFINE=$1 DEFTET_PIT_XFINE=$2 DEFTET_PIT_QDIV=$3 python tools/probes/scan_variants.py --config $c --algo 0 --tet-order native --reps 10 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('yz=$1 x=$2 qdiv=$3', r['config'], r['grid_yz_x'], r['traversal_us_in_step'], r['step_us'], r['fwd_us_warm'])"
  done
done
