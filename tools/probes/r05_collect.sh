#!/bin/bash
# After `gpurun -- tools/probes/r05_evidence.sh …` has merged its files into gpurun_out/r05/: copy what is judged into profiles/ and
# rewrite the README rows that quote it.   tools/probes/r05_collect.sh
cd "$(dirname "$0")/../.."
for f in r05_bench_kernel_stats.csv r05_bench_line.json r05_bench_line_driver_style.jsonl r05_bench_line_measured_box.json r05_bench_line_under_rocprof.json \
         r05_bench_lines_raster_geometry.jsonl r05_geometry_step_kernel_stats.csv r05_geometry_step_line_under_rocprof.json r05_pmc_geometry.json r05_pmc_raster.json \
         r05_pmc_traffic.json r05_pmc_traversal.json r05_raster_step_kernel_stats.csv r05_raster_step_line_under_rocprof.json r05_tolerances.jsonl; do
  [ gpurun_out/r05/$f -nt profiles/$f ] && cp gpurun_out/r05/$f profiles/$f && echo "updated $f"
done
python tools/profiles_readme_rows.py
