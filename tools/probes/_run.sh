#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_surface_ops_gpu.py -q -x -k "chamfer" 2>&1 | tail -15
