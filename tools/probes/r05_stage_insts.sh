#!/bin/bash
# VALU / SALU / LDS wave-instructions of k_tet_scan_wave per stage: probe builds that stop after stage n (tools/probes/build_variant.sh ... -DPIT_STOP=n)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for n in ${STAGES:-1 2 3 4 5 0}; do
  lib=""; [ $n != 0 ] && lib="DEFTET_HIP_LIB=$PWD/tools/probes/bin/libdeftet_stop$n.so"
  env $lib PMC_PASSES="0 6" tools/pmc_run.sh gpurun_out/stage$n.json k_tet_scan_wave -- python $PWD/tools/probes/scan_variants.py --config ${CFG:-2} --algo 4 --tet-order native --reps 3 > /dev/null 2>&1
  python - <<PY
import json
r=json.load(open("gpurun_out/stage$n.json"))
for k,v in r.items():
    print("stop=$n", {c: v.get(c) for c in ("SQ_WAVES","SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_VMEM_RD","SQ_INSTS_VMEM_WR","SQ_INSTS_SMEM")})
PY
done
