#!/usr/bin/env python
"""Rewrites the rows of profiles/README.md that quote numbers of the evidence pass (bench lines, kernel tables, pmc summaries) from the
files themselves, so that a refreshed pass cannot leave a stale figure behind:  python tools/profiles_readme_rows.py [--check]"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def jl(name):
    return [json.loads(l) for l in open(os.path.join(P, name)) if l.strip()]


def kstat(name, pat):
    for r in csv.DictReader(open(os.path.join(P, name))):
        if pat in r["Name"]:
            return float(r["AverageNs"]) / 1000, int(r["Calls"])
    raise KeyError(pat)


d = jl("r05_bench_line.json")[0]
r, cb, bf = d["roofline"], d["cpu_baseline"], d["brute_force_hip"]
oc = {o["config_id"]: o for o in d["other_configs"]}
drv = jl("r05_bench_line_driver_style.jsonl")
mb = jl("r05_bench_line_measured_box.json")[0]
ur = jl("r05_bench_line_under_rocprof.json")[0]
rg = jl("r05_bench_lines_raster_geometry.jsonl")
tv = json.load(open(os.path.join(P, "r05_pmc_traversal.json")))["deftet::pit::k_tet_scan_wave<false>"]
tr = json.load(open(os.path.join(P, "r05_pmc_traffic.json")))
M = 1e6
rows = {
    "r05_bench_line.json": "| `r05_bench_line.json` | `python bench.py` (default: configs[2], 20 steps, 5 warm-up) | the driver's line: **`ms_per_step` %.4f** (median %.4f, max %.4f), `roofline` (traversal `k_tet_scan_wave` %.5f ms per launch, **frac %.4f**; `traffic` %.1f MB from `r05_pmc_traffic.json`; `peak_measured` %.2f TB/s copy / %.2f read), `ms_per_step_pipelined` %.4f, `ms_per_step_hipgraph` %.4f, `brute_force_hip` (%.0f ms for the forward of all eight shapes: %s×, same result), `cpu_baseline` (%s M/s on %d threads of an EPYC 9575F, %.0f M/s on one core; `covers: fwd`, `value_fwd_bwd` %s), `config.tet_order` / `config.query_box` (%s), `other_configs`: configs[1] %.4f ms (median %.4f), configs[3] %.4f ms (frac %.3f), rasterizer %.3f ms (`roofline.valu.frac` %.2f), geometry step %.3f ms (median %.3f; `roofline.bound` \"valu\", frac %.2f) |" % (
        d["ms_per_step"], d["ms_per_step_median"], d["ms_per_step_max"], r["avg_launch_ms"], r["frac"], r["traffic"] / M, r["peak_measured"]["copy_GBs"] / 1e3, r["peak_measured"]["read_GBs"] / 1e3,
        d["ms_per_step_pipelined"], d["ms_per_step_hipgraph"], bf["ms_fwd_batch"], format(round(bf["speedup_of_binned_fwd_bwd_step_over_brute_fwd"]), ","),
        format(round(cb["value"]), ","), cb["cores"], cb["value_1core"], format(round(cb["value_fwd_bwd"]), ","),
        re.search(r"calls so far: (.*)$", d["config"]["query_box"]).group(1).replace("fall-backs to measuring", "fall-backs"),
        oc[1]["ms_per_step"], oc[1]["ms_per_step_median"], oc[3]["ms_per_step"], oc[3]["roofline"]["frac"], oc[4]["ms_per_step"], oc[4]["roofline"]["valu"]["frac"],
        oc[5]["ms_per_step"], oc[5]["ms_per_step_median"], oc[5]["roofline"]["frac"]),
    "r05_bench_line_driver_style.jsonl": "| `r05_bench_line_driver_style.jsonl` | `python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-brute-force`, three times | the command the driver runs: **%s ms per step**, traversal %s ms per launch = frac %s; hipGraph replay %.3f–%.3f |" % (
        " / ".join("%.4f" % x["ms_per_step"] for x in drv), " / ".join("%.5f" % x["roofline"]["avg_launch_ms"] for x in drv), " / ".join("%.3f" % x["roofline"]["frac"] for x in drv),
        min(x["ms_per_step_hipgraph"] for x in drv), max(x["ms_per_step_hipgraph"] for x in drv)),
    "r05_bench_line_measured_box.json": "| `r05_bench_line_measured_box.json` | the same with `DEFTET_BENCH_QUERY_BOX=measure` (every step measures its own query box: the round-4 launch sequence) | %.4f ms per step against %.4f (mean of the three above) with the tracked box |" % (
        mb["ms_per_step"], sum(x["ms_per_step"] for x in drv) / len(drv)),
    "r05_bench_kernel_stats.csv": "| `r05_bench_kernel_stats.csv`, `r05_bench_line_under_rocprof.json` | `rocprofv3 --kernel-trace --stats --output-format csv … -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-brute-force --no-bandwidth-probe` | per-kernel averages under the profiler: `k_tet_scan_wave<false>` **%.1f µs** (%d launches), `k_bary_bwd_hits` %.1f, `k_finalize` %.1f, `k_slab_sort` %.1f, `k_slab_local<true>` %.1f, `k_rowdot_fused` %.1f; one-time launches of the process: the order choice (`k_tet_scan_wave<true>` × 4, `k_order_*`), the first call's `k_query_bbox` / `k_slab_local<false>`, the two-launch row dots of the captured graphs; the line the same process printed: %.4f ms, %.4f ms per launch by the library's own events |" % (
        kstat("r05_bench_kernel_stats.csv", "k_tet_scan_wave<false>") + (kstat("r05_bench_kernel_stats.csv", "k_bary_bwd_hits")[0], kstat("r05_bench_kernel_stats.csv", "k_finalize")[0],
        kstat("r05_bench_kernel_stats.csv", "k_slab_sort")[0], kstat("r05_bench_kernel_stats.csv", "k_slab_local<true>")[0], kstat("r05_bench_kernel_stats.csv", "k_rowdot_fused<false>")[0],
        ur["ms_per_step"], ur["roofline"]["avg_launch_ms"])),
    "r05_pmc_traversal.json": "| `r05_pmc_traversal.json` | `tools/pmc_run.sh … k_tet_scan -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-bandwidth-probe --no-brute-force` (seven `--pmc` passes, each in its own run with `--kernel-trace` only) | per-launch averages for `k_tet_scan_wave<false>`: **`SQ_INSTS_VALU` %.2f M** (round 4: 26.9 M), `SQ_INSTS_SALU` %.1f M, `SQ_INSTS_LDS` %.2f M, `TCP_TOTAL_CACHE_ACCESSES` %.1f M, `TA_TA_BUSY` %.1f M, `TCC_HIT_sum` %.2f M / `TCC_MISS_sum` %.2f M / `TCC_REQ_sum` %.2f M, `FETCH_SIZE` ×2 %.1f MB, `WRITE_SIZE` %.1f MB; `<true>` = the four launches of the order choice |" % (
        tv["SQ_INSTS_VALU"] / M, tv["SQ_INSTS_SALU"] / M, tv["SQ_INSTS_LDS"] / M, tv["TCP_TOTAL_CACHE_ACCESSES"] / M, tv["TA_TA_BUSY"] / M, tv["TCC_HIT_sum"] / M, tv["TCC_MISS_sum"] / M,
        tv["TCC_REQ_sum"] / M, tv["FETCH_SIZE_bytes_x2"] / M, tv["WRITE_SIZE_bytes"] / M),
    "r05_pmc_traffic.json": "| `r05_pmc_traffic.json` | same runs, `tools/pmc_traffic.py` on the FETCH_SIZE / WRITE_SIZE passes | HBM-side bytes per launch of every kernel; **`whole_step_hbm_bytes` %.3f GB** = the six kernels launched once per step (`step_kernels`; the one-time launches are listed with their launch counts and not summed); `commit` `%s`, `kernel_source_sha1` |" % (
        tr["whole_step_hbm_bytes"] / 1e9, tr["commit"][:7]),
    "r05_bench_lines_raster_geometry.jsonl": "| `r05_bench_lines_raster_geometry.jsonl` | `python bench.py --config {4,5} --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-bandwidth-probe` | rasterizer **%.3f ms** fwd+bwd (round 4: 2.42), geometry step **%.3f ms** (median %.3f; %.3f inside the default line; 2.43–2.50 on the other boxes), both with `roofline.valu` |" % (
        rg[0]["ms_per_step"], rg[1]["ms_per_step"], rg[1]["ms_per_step_median"], oc[5]["ms_per_step"]),
}
path = os.path.join(P, "README.md")
lines = open(path).read().split("\n")
out, done = [], set()
for ln in lines:
    hit = [k for k in rows if ln.startswith("| `%s`" % k)]
    if hit:
        out.append(rows[hit[0]]); done.add(hit[0])
    else:
        out.append(ln)
missing = set(rows) - done
assert not missing, missing
new = "\n".join(out)
if "--check" in sys.argv:
    sys.exit(0 if new == open(path).read() else "profiles/README.md quotes figures that are not in the files: run tools/profiles_readme_rows.py")
open(path, "w").write(new)
print("rewrote %d rows" % len(rows))
