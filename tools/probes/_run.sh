#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r3h; mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline > $O/bench_graph.json 2> $O/bench_graph_err.txt; tail -3 $O/bench_graph_err.txt
python - <<'PY'
import json
b=json.load(open('gpurun_out/r3h/bench_graph.json'))
print(b['ms_per_step'], b.get('ms_per_step_hipgraph'), b.get('hipgraph_note'))
for o in b['other_configs']: print(o['config_id'], o['ms_per_step'], o.get('ms_per_step_hipgraph'))
PY
