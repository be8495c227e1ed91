#!/bin/bash
# Counter passes (rocprofv3 --pmc, each in its own run with --kernel-trace only) over one command; per-kernel averages
# are written as JSON by tools/pmc_summary.py.
#   tools/pmc_run.sh <out.json> <kernel substring> -- <command ...>
# Passes: SQ instruction / cycle counters, SQ activity / wait counters, TA/TCP (vector memory path), TCC (L2 hit / miss /
# requests), FETCH_SIZE, WRITE_SIZE (separate passes as MI355X_MICROARCH.md prescribes).
set -e
out=$1; match=$2; shift 3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$(basename "$out" .json)
cd /tmp && export TMPDIR=/tmp
passes=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
 "TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"
)
if [ -n "${PMC_PASSES:-}" ]; then sel=(); for i in $PMC_PASSES; do sel+=("${passes[$i]}"); done; passes=("${sel[@]}"); fi
dirs=()
i=0
for p in "${passes[@]}"; do
    d=$R/gpurun_out/pmc_${tag}_$i
    rm -rf "$d"
    rocprofv3 --pmc $p --kernel-trace -d "$d" -o p --output-format csv -- "$@" > "$R/gpurun_out/pmc_${tag}_$i.log" 2>&1 || echo "pass $i failed (see gpurun_out/pmc_${tag}_$i.log)"
    dirs+=("$d")
    i=$((i+1))
done
python $R/tools/pmc_summary.py "${dirs[@]}" --match "$match" > "$R/$out"
# PMC_TRAFFIC_OUT=<file>: also the per-kernel HBM traffic table bench.py reads (FETCH_SIZE / WRITE_SIZE passes, all kernels)
if [ -n "${PMC_TRAFFIC_OUT:-}" ] && [ -z "${PMC_PASSES:-}" ]; then python $R/tools/pmc_traffic.py "${dirs[4]}" "${dirs[5]}" > "$R/$PMC_TRAFFIC_OUT"; fi
for d in "${dirs[@]}"; do rm -rf "$d"; done      # the raw per-dispatch CSVs are large; the summary is what is kept
echo "wrote $out"
