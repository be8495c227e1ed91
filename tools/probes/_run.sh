#!/bin/bash
# scratch GPU script of the current experiment (overwritten freely)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r3f; mkdir -p $O
timeout 600 python -m pytest tests -q -m gpu -x -k "energ or tet_ops or pipeline or deftet_module" 2>&1 | tail -3
timeout 300 python tools/probes/energy_probe.py | tee $O/energy_probe.json
timeout 300 python tools/probes/energy_probe.py --sets 1 | tee $O/energy_probe_warm.json
timeout 600 rocprofv3 --kernel-trace --stats -d $O/energy_prof -o e --output-format csv -- python tools/probes/energy_probe.py > /dev/null 2>&1
find $O/energy_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/energy_kernel_stats.csv; rm -rf $O/energy_prof
head -4 $O/energy_kernel_stats.csv | cut -c1-200
