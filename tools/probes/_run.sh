#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_pipeline_gpu.py tests/test_surface_ops_gpu.py tests/test_vertex_ops_gpu.py -q -x 2>&1 | tail -5
for i in 1 2; do timeout 300 python bench.py --config 5 --no-cpu-baseline --no-other-configs --no-bandwidth-probe 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print(b['ms_per_step'])"; done
timeout 600 python tools/bench_ops.py 2>/dev/null | grep "nn_index" | cut -c1-200
