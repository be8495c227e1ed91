#!/bin/bash
# A/B of nontemporal tet loads (probe builds -DPIT_SCAN_TET_NT=1 / -DPIT_BWD_TET_NT=1): the step's kernels by the library's own events
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/r05b
for rep in 1 2; do
for n in ${VARIANTS:-product scannt bwdnt bothnt}; do
  lib=""; [ "$n" != product ] && lib="DEFTET_HIP_LIB=$PWD/tools/probes/bin/libdeftet_$n.so"
  for c in 2 3; do
    echo -n "$n " ; env $lib python tools/probes/sort_probe.py --config $c 2>/dev/null | tail -1
  done
done
done | tee gpurun_out/r05b/${OUT:-nt_loads}.txt
