#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r3i; mkdir -p $O
timeout 900 python -m pytest tests/test_surface_ops_gpu.py tests/test_pipeline_gpu.py -q -x 2>&1 | tail -5

