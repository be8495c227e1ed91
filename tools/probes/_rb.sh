cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_raster_gpu.py -m gpu -x -q 2>&1 | tail -4
for o in sorted runs sorted runs; do
  DEFTET_RAST_BWD=$o python tools/probes/raster_kernels_probe.py 0 2>/dev/null | grep '^{' | sed "s/^/$o /"
done
python tools/probes/raster_determinism_probe.py 2>&1 | tail -2
