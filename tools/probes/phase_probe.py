#!/usr/bin/env python
"""Per-phase wall cycles of the traversal kernels (s_memtime deltas summed over waves), from a diagnostic build:
    tools/probes/build_variant.sh tools/probes/bin/libdeftet_phase.so -DPIT_PHASE_TIMING
    DEFTET_HIP_LIB=$PWD/tools/probes/bin/libdeftet_phase.so python tools/probes/phase_probe.py [--config 2]
Prints average cycles per wave and per phase for algo 0 (k_tet_scan_fma) and 10 (k_tet_scan_lds<true>)."""
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
a = ap.parse_args()
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.deftet_debug_phase_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
dev = torch.device("cuda:0")
wl = bench.PitWorkload(dict(bench.CONFIGS[a.config], sets=1), 0, dev, 1, None, pipeline=False)
d = wl.sets[0]
names = {0: ["load+setup", "loop", "publish"], 10: ["load+setup", "union box", "cell starts->LDS", "offset scan + queries->LDS", "loop", "publish"],
         9: ["load+setup", "union box", "cell starts->LDS", "(no query staging)", "loop", "publish"]}
base = {0: 0, 9: 4, 10: 4}
for algo in (0, 9, 10):
    for _ in range(2):
        hip_ops.point_in_tet(d["tet"], d["pts"], want_bary=True, pred_bxt=d["pred"], want_hits=True, algo=algo)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    raw.deftet_debug_phase_read(buf, 1)
    reps = 4
    for _ in range(reps):
        hip_ops.point_in_tet(d["tet"], d["pts"], want_bary=True, pred_bxt=d["pred"], want_hits=True, algo=algo)
    torch.cuda.synchronize()
    raw.deftet_debug_phase_read(buf, 1)
    n_waves = wl.B * ((wl.T + 63) // 64) * reps
    vals = [buf[base[algo] + i] / n_waves for i in range(len(names[algo]))]
    print(json.dumps({"algo": algo, "kernel": hip_ops.pit_kernel_name(algo), "waves_per_launch": n_waves // reps,
                      "cycles_per_wave": {k: round(v) for k, v in zip(names[algo], vals)}, "total": round(sum(vals))}), flush=True)
