#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_raster_gpu.py -q -x 2>&1 | tail -4
for pol in 0 1; do
DEFTET_BENCH_RASTER_POLICY=$pol timeout 300 python bench.py --config 4 --no-cpu-baseline --no-other-configs --no-bandwidth-probe 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print(b['ms_per_step'], b['roofline']['kernel'], b['roofline']['avg_launch_ms'])"
done
