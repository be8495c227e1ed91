#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for p in 0 1 0 1; do python tools/probes/raster_kernels_probe.py $p 2>&1 | grep -v amdgpu.ids | tail -1; done
timeout 900 python -m pytest tests/test_raster_gpu.py -x -q 2>&1 | grep -v amdgpu | tail -3
