#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_point_in_tet_gpu.py tests/test_fuzz_gpu.py -q -x 2>&1 | tail -2
for rep in 1 2; do for c in 2 3; do
DEFTET_HIP_LIB=$GRAFT_REPO_ROOT/tools/probes/bin/libdeftet_head.so python tools/probes/scan_variants.py --config $c --reps 30 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('head', r['config'], r['traversal_us_in_step'], r['step_us'], r['bwd_us_warm'])"
python tools/probes/scan_variants.py --config $c --reps 30 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('new ', r['config'], r['traversal_us_in_step'], r['step_us'], r['bwd_us_warm'])"
done; done
