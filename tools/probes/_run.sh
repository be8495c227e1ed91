#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r3n; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
for pol in 0 1; do
DEFTET_BENCH_RASTER_POLICY=$pol timeout 300 python bench.py --config 4 --no-cpu-baseline --no-other-configs --no-bandwidth-probe 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('raster', b['ms_per_step'], b['roofline']['avg_launch_ms'])"
done
timeout 300 python bench.py --config 5 --no-cpu-baseline --no-other-configs --no-bandwidth-probe 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('geometry', b['ms_per_step'])"
timeout 900 python tools/bench_ops.py > $O/bench_ops.jsonl 2>/dev/null; grep "nn_index\|sparse_render\|tri_dist_bwd" $O/bench_ops.jsonl | cut -c1-220
