#!/bin/bash
# Builds an alternative libdeftet_hip.so with extra -D flags for A/B runs (load it with DEFTET_HIP_LIB=<path>).
#   tools/probes/build_variant.sh tools/probes/bin/libdeftet_w8.so -DPIT_WAVES=8
#   tools/probes/build_variant.sh tools/probes/bin/libdeftet_x.so --pit-only -DPIT_BATCH=3    # 0.3 MB instead of 20 MB
set -e
out=$1; shift
cd "$(dirname "$0")/../.."
args=()
for a in "$@"; do
    if [ "$a" = "--pit-only" ]; then args+=(--only point_in_tet.hip,common.cpp,reduce.hip,probe.hip,tet_order.hip,vertex_ops.hip,prims.hip)    # small library: point-in-tet + row dots only
    else args+=("$a"); fi
done
python -m deftet_amd.build --out "$out" "${args[@]}" | tail -1
