#!/bin/bash
# scratch GPU script of the current experiment (overwritten freely)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r3g; mkdir -p $O
R=$GRAFT_REPO_ROOT
PMC_TRAFFIC_OUT=$O/pmc_traffic.json bash tools/pmc_run.sh $O/pmc_traversal.json k_tet_scan_slab -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-bandwidth-probe
cd $R
cp $O/pmc_traffic.json profiles/r03_pmc_traffic.json
timeout 400 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; cut -c1-1500 $O/bench_line.json
cat $O/pmc_traversal.json | head -60
