#!/bin/bash
# -fno-slp-vectorize per source file: bench config 4 (rasterizer) / 5 (geometry step) with probe builds of the library
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
run() {  # lib config
  lib=""; [ "$1" != product ] && lib="DEFTET_HIP_LIB=$PWD/tools/probes/bin/libdeftet_$1.so"
  env $lib python bench.py --config $2 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-bandwidth-probe 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 config $2:', r['ms_per_step'], r['ms_per_step_median'], r['roofline']['kernel'], r['roofline']['avg_launch_ms'])"
}
run product 4; run noslp_raster 4; run product 4; run noslp_raster 4
run product 5; run noslp_surface_ops 5; run noslp_tet_ops 5; run noslp_check_sign 5; run noslp_vertex_ops 5; run product 5; run noslp_surface_ops 5
