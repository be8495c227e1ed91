#!/bin/bash
# Round 5: wave-vs-slab traversal on three tet orders (Kuhn enumeration, the same list shuffled, the shipped QuarTet grid), each
# with the caller's numbering and with the computed column order -> gpurun_out/r05_scan_ab_orders.jsonl
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_scan_ab_orders.jsonl
rm -f $out
run() {  # config mesh order algo
    m=""; [ "$2" != "kuhn" ] && m="--mesh $2"
    timeout 300 python tools/probes/scan_variants.py --config $1 $m --tet-order $3 --algo $4 --reps 10 >> $out 2>> gpurun_out/r05_orders_err.log
}
for algo in 4 3; do
    run 2 kuhn native $algo
    run 2 kuhn sorted $algo
    run 2 shuffled native $algo
    run 2 shuffled sorted $algo
    run 1 cube40 native $algo
    run 1 cube40 sorted $algo
done
run 2 shuffled auto 0
run 2 kuhn auto 0
run 1 cube40 auto 0
run 3 kuhn native 4
run 3 shuffled sorted 4
cat $out | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print(r['config'], r['mesh'], r['tet_order'], r['kernel'], r['traversal_us_in_step'], r['step_us'], r['stats_irrT_irrQ_ovf_x_x_rescanned_ovfTets'])
"
