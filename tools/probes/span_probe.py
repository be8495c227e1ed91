#!/usr/bin/env python
"""Occupancy of one k_tet_scan_wave launch over time, from the start / end wall-clock stamps of every wave (probe build
-DPIT_PHASE_TIMING): DEFTET_HIP_LIB=tools/probes/bin/libdeftet_phase.so python tools/probes/span_probe.py [--config 2] [--algo 4]"""
import argparse, ctypes, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deftet_amd import _lib, hip_ops  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--config", type=int, default=2); ap.add_argument("--algo", type=int, default=4)
a = ap.parse_args()
lib = _lib.load(); raw = ctypes.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda:0")
wl = bench.PitWorkload(dict(bench.CONFIGS[a.config], sets=1, tet_order="native", query_box="measure"), 0, dev, 1, None, pipeline=False)
d = wl.sets[0]
for _ in range(3):
    hip_ops.point_in_tet(d["tet"], d["pts"], want_bary=True, pred_bxt=d["pred"], want_hits=True, algo=a.algo)
torch.cuda.synchronize()
per = 2 if a.algo == 5 else 1
WG = int(os.environ.get("PIT_WG", "128"))          # threads per workgroup of the build under test
nblk = (((wl.T + per - 1) // per + WG - 1) // WG + 7) // 8 * 8
n = wl.B * nblk * 4                                  # (the probe numbers its rows 4 per workgroup whatever the workgroup size)
buf = (ctypes.c_ulonglong * (2 * n))()
assert raw.deftet_debug_span_read(buf, n) == 0
sp = np.frombuffer(buf, dtype=np.uint64).reshape(n, 2).astype(np.int64)
sp = sp[(np.arange(n) % 4) < WG // 64]               # the rows this kernel's waves write (k_slab_sort stamps the same table before it)
sp = sp[(sp[:, 1] > sp[:, 0])]                      # waves that ran to the end (a block's padding waves return early)
sp = sp[sp[:, 0] >= np.percentile(sp[:, 0], 1) - 500]   # (rows of padding workgroups still hold k_slab_sort's earlier stamps: > 5 us before the launch)
t0 = sp[:, 0].min(); sp = (sp - t0) * 10e-3         # 100 MHz ticks -> microseconds
life = sp[:, 1] - sp[:, 0]
end = sp[:, 1].max()
edges = np.linspace(0, end, 41)
occ = [float(((sp[:, 0] < hi) & (sp[:, 1] > lo)).sum()) / (1024.0) for lo, hi in zip(edges[:-1], edges[1:])]   # waves per SIMD overlapping the bin
print(json.dumps({"config": a.config, "algo": a.algo, "waves": int(sp.shape[0]), "kernel_us_first_start_to_last_end": round(float(end), 2),
                  "wave_life_us": {"mean": round(float(life.mean()), 2), "p50": round(float(np.median(life)), 2), "p90": round(float(np.percentile(life, 90)), 2),
                                   "p99": round(float(np.percentile(life, 99)), 2), "max": round(float(life.max()), 2)},
                  "last_start_us": round(float(sp[:, 0].max()), 2),
                  "waves_per_simd_in_40_time_bins": [round(x, 1) for x in occ]}))
