#!/bin/bash
# Round 6 evidence pass (one box, once): everything lands in gpurun_out/r06/, the files that are judged are then copied to profiles/.
#   gpurun -- 'DEFTET_COMMIT=<git rev-parse HEAD> tools/probes/r06_evidence.sh [part ...]'     parts: pmc bench prof ab host tol (default: all)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
parts="${*:-pmc bench prof ab host tol}"
B="python $PWD/bench.py"
Q="--no-cpu-baseline --no-other-configs --no-bandwidth-probe --no-brute-force"
has() { case " $parts " in *" $1 "*) return 0;; esac; return 1; }

if has pmc; then
  PMC_TRAFFIC_OUT=$O/r06_pmc_traffic.json tools/pmc_run.sh $O/r06_pmc_traversal.json k_tet_scan -- $B --steps 5 --warmup 2 $Q
  PMC_PASSES="0 1 4 5" tools/pmc_run.sh $O/r06_pmc_raster.json k_pix_raster -- $B --config 4 --steps 3 --warmup 1 $Q
  PMC_PASSES="0 1" tools/pmc_run.sh $O/r06_pmc_geometry.json k_tri_query_coop -- $B --config 5 --steps 2 --warmup 1 $Q
  cp $O/r06_pmc_traffic.json $O/r06_pmc_raster.json $O/r06_pmc_geometry.json profiles/     # bench.py quotes traffic / VALU counts from profiles/
fi
if has bench; then
  $B 2> $O/bench_default.err | tail -1 > $O/r06_bench_line.json
  rm -f $O/r06_bench_line_driver_style.jsonl
  for i in 1 2 3; do $B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-brute-force 2>/dev/null | tail -1 >> $O/r06_bench_line_driver_style.jsonl; done
  DEFTET_BENCH_QUERY_BOX=measure $B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-brute-force 2>/dev/null | tail -1 > $O/r06_bench_line_measured_box.json
  rm -f $O/r06_bench_lines_raster_geometry.jsonl
  for c in 4 5; do $B --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-bandwidth-probe 2>/dev/null | tail -1 >> $O/r06_bench_lines_raster_geometry.jsonl; done
  DEFTET_BENCH_RASTER_POLICY=1 $B --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-bandwidth-probe 2>/dev/null | tail -1 >> $O/r06_bench_lines_raster_geometry.jsonl
fi
if has prof; then
  for spec in "2 r06_bench" "4 r06_raster_step" "5 r06_geometry_step"; do
    set -- $spec
    rm -rf $O/prof_$1
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$1 -o p -- $B --config $1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-brute-force --no-bandwidth-probe 2>/dev/null | tail -1 > $GRAFT_REPO_ROOT/$O/$2_line_under_rocprof.json)
    f=$(find $O/prof_$1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/$2_kernel_stats.csv
    rm -rf $O/prof_$1
  done
fi
if has ab; then
  rm -f $O/r06_scan_ab_kernels.jsonl
  for c in 2 3 1; do for a in 4 3; do python tools/probes/scan_variants.py --config $c --algo $a --tet-order native --reps 30 2>/dev/null >> $O/r06_scan_ab_kernels.jsonl; done; done
  python tools/probes/scan_variants.py --config 2 --algo 4 --mesh shuffled --tet-order auto --reps 30 2>/dev/null >> $O/r06_scan_ab_kernels.jsonl
  python tools/probes/scan_variants.py --config 1 --algo 0 --mesh cube40 --tet-order auto --reps 30 2>/dev/null >> $O/r06_scan_ab_kernels.jsonl
  rm -f $O/r06_step_kernels.jsonl
  for c in 2 1 3; do python tools/probes/sort_probe.py --config $c 2>/dev/null | tail -1 >> $O/r06_step_kernels.jsonl; done
  rm -f $O/r06_raster_kernels.jsonl
  for p in 0 1; do python tools/probes/raster_kernels_probe.py $p 2>/dev/null | tail -1 >> $O/r06_raster_kernels.jsonl; done
fi
if has host; then
  python tools/probes/bwd_to_vertices_probe.py 2>/dev/null | grep '^{' > $O/r06_bwd_to_vertices.jsonl
  python tools/probes/geometry_cpu_probe.py 2>/dev/null | grep '^{' > $O/r06_geometry_host_time.json
  python tools/probes/host_call_probe.py 2>/dev/null | grep '^{' > $O/r06_host_call_cost.json
  DEFTET_STEP_FUSED_BWD=0 $B --config 5 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-bandwidth-probe 2>/dev/null | tail -1 > $O/r06_geometry_line_two_stage_bwd.json
fi
if has tol; then
  rm -f $O/r06_tolerances.jsonl
  DEFTET_TOLERANCE_REPORT=$PWD/$O/r06_tolerances.jsonl timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
  grep -v amdgpu.ids $O/pytest_gpu.log | tail -3
  grep -E '[0-9]+ passed' $O/pytest_gpu.log | tail -1 > $O/r06_pytest_gpu.txt
fi
ls -la $O
