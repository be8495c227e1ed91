#!/usr/bin/env python
"""Per-phase wall cycles of the traversal kernels (s_memtime deltas summed over waves), from a diagnostic build:
    tools/probes/build_variant.sh tools/probes/bin/libdeftet_phase.so -DPIT_PHASE_TIMING
    DEFTET_HIP_LIB=$PWD/tools/probes/bin/libdeftet_phase.so python tools/probes/phase_probe.py [--config 2]
Prints average cycles per wave and per phase of the default traversal kernel."""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deftet_amd import _lib, hip_ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--algo", type=int, default=0)
a = ap.parse_args()
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.deftet_debug_phase_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
dev = torch.device("cuda:0")
wl = bench.PitWorkload(dict(bench.CONFIGS[a.config], sets=1), 0, dev, 1, None, pipeline=False)
d = wl.sets[0]
names = ["load+setup", "loop", "publish"] if a.algo == 3 else ["setup+groups", "staged(rest)", "global walk", "exact+records", "bounds+scan", "owners+copy", "traverse", "sum_Nc", "s8", "s9", "s10", "s11", "s12", "lane0_cands", "fallback_lanes", "chunks"]
for _ in range(2):
    hip_ops.point_in_tet(d["tet"], d["pts"], want_bary=True, pred_bxt=d["pred"], want_hits=True, algo=a.algo)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
raw.deftet_debug_phase_read(buf, 1)
rbuf = (ctypes.c_ulonglong * 8)()
if hasattr(raw, "deftet_debug_reason_read"):
    raw.deftet_debug_reason_read(rbuf, 1)
reps = 4
for _ in range(reps):
    hip_ops.point_in_tet(d["tet"], d["pts"], want_bary=True, pred_bxt=d["pred"], want_hits=True, algo=a.algo)
torch.cuda.synchronize()
raw.deftet_debug_phase_read(buf, 1)
n_waves = wl.B * ((wl.T + 63) // 64) * reps
vals = [buf[i] / n_waves for i in range(len(names))]
G = hip_ops.point_in_tet_grid(wl.T, wl.Q)[0]
n_sort = wl.B * G * 4 * 4 * reps                                    # k_slab_sort: 4 waves per (slab quarter, shape) workgroup
print(json.dumps({"kernel": "k_slab_sort", "waves_per_launch": n_sort // reps, "cycles_per_wave": {
    k: round(buf[8 + i] / n_sort) for i, k in enumerate(["loads+clear", "count", "scan", "placement", "table"])}}), flush=True)
print(json.dumps({"kernel": hip_ops.pit_kernel_name(a.algo, wl.T, wl.Q), "waves_per_launch": n_waves // reps,
                  "cycles_per_wave": {k: round(v, 2) for k, v in zip(names, vals)}, "total": round(sum(vals[:7]))}), flush=True)
if hasattr(raw, "deftet_debug_reason_read"):
    raw.deftet_debug_reason_read(rbuf, 1)
    print(json.dumps({"ungrouped_lanes_per_launch": dict(zip(["few_left_p0", "few_left_p1", "small_pivot_group(all remaining)", "five_rows", "slabs>64", "rows>cap",
                                                               "left_after_two_groups", "row_longer_than_chunk"], [int(rbuf[i]) // reps for i in range(8)]))}), flush=True)
