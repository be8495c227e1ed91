#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r3k; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.txt
python - <<'PY'
import json
b=json.load(open('gpurun_out/r3k/bench_line.json'))
print(b['ms_per_step'], b['roofline']['frac'], b['roofline']['avg_launch_ms'], b.get('ms_per_step_hipgraph'))
for o in b['other_configs']: print(o['config_id'], o['ms_per_step'], o.get('ms_per_step_hipgraph'))
PY
