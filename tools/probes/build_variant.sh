#!/bin/bash
# Builds an alternative libdeftet_hip.so with extra -D flags for A/B runs (load it with DEFTET_HIP_LIB=<path>).
#   tools/probes/build_variant.sh tools/probes/bin/libdeftet_w8.so -DPIT_WAVES=8
set -e
out=$1; shift
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wno-unused-function "$@" \
    -x hip deftet_amd/csrc/*.hip deftet_amd/csrc/*.cpp -o "$out"
echo "built $out"
