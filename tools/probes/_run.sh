#!/bin/bash
# scratch GPU script of the current experiment (overwritten freely)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r3p; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
PMC_TRAFFIC_OUT=$O/pmc_traffic.json bash tools/pmc_run.sh $O/pmc_traversal.json k_tet_scan_slab -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-bandwidth-probe
cd $R
cp $O/pmc_traffic.json profiles/r03_pmc_traffic.json
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o b --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs > $O/bench_line_under_rocprof.json 2> $O/err.txt
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv; rm -rf $O/prof
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.txt
python - <<'PY'
import json
b=json.load(open('gpurun_out/r3p/bench_line.json'))
print(b['ms_per_step'], b['roofline']['frac'], b['roofline']['avg_launch_ms'], b['roofline']['traffic'], b['roofline']['traffic_commit'], b.get('ms_per_step_hipgraph'))
for o in b['other_configs']: print(o['config_id'], o['ms_per_step'], o.get('ms_per_step_hipgraph'))
PY
for rep in 1 2; do
  DEFTET_HIP_LIB=$R/tools/probes/bin/libdeftet_r02.so python tools/probes/scan_variants.py --reps 30 --kernel 'k_tet_scan_fma<false>' >> $O/scan_ab_vs_r02.jsonl 2>/dev/null
  python tools/probes/scan_variants.py --reps 30 >> $O/scan_ab_vs_r02.jsonl 2>/dev/null
done
for c in 1 3; do
  DEFTET_HIP_LIB=$R/tools/probes/bin/libdeftet_r02.so python tools/probes/scan_variants.py --config $c --reps 20 --kernel 'k_tet_scan_fma<false>' >> $O/scan_ab_vs_r02.jsonl 2>/dev/null
  python tools/probes/scan_variants.py --config $c --reps 20 >> $O/scan_ab_vs_r02.jsonl 2>/dev/null
done
