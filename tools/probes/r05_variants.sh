#!/bin/bash
# times the traversal of several probe builds of the library (tools/probes/bin/libdeftet_<name>.so) at configs[2] and [3]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for n in "$@"; do
  for c in 2 3; do
    lib=""; [ "$n" != product ] && lib="DEFTET_HIP_LIB=$PWD/tools/probes/bin/libdeftet_$n.so"
    env $lib python tools/probes/scan_variants.py --config $c --algo 4 --tet-order native --reps 10 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$n', r['config'], r['kernel'], r['traversal_us_in_step'], r['step_us'], r['traversal_us_warm'])"
  done
done
