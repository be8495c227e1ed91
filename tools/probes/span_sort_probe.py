#!/usr/bin/env python
"""Occupancy of one k_slab_sort launch over time (probe build -DPIT_PHASE_TIMING): start / end wall-clock stamps of every wave."""
import argparse, ctypes, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deftet_amd import _lib, hip_ops  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--config", type=int, default=2)
a = ap.parse_args()
lib = _lib.load(); raw = ctypes.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda:0")
wl = bench.PitWorkload(dict(bench.CONFIGS[a.config], sets=1, tet_order="native", query_box="measure"), 0, dev, 1, None, pipeline=False)
d = wl.sets[0]
for _ in range(3):
    pq = hip_ops.prepare_queries(d["pts"], wl.T)
torch.cuda.synchronize()
G = hip_ops.point_in_tet_grid(wl.T, wl.Q)[0]
n = wl.B * G * 4 * 4
buf = (ctypes.c_ulonglong * (2 * n))()
assert raw.deftet_debug_span_read(buf, n) == 0
sp = np.frombuffer(buf, dtype=np.uint64).reshape(n, 2).astype(np.int64)
sp = sp[(sp[:, 1] > sp[:, 0])]
t0 = sp[:, 0].min(); sp = (sp - t0) * 10e-3
life = sp[:, 1] - sp[:, 0]; end = sp[:, 1].max()
edges = np.linspace(0, end, 21)
occ = [float(((sp[:, 0] < hi) & (sp[:, 1] > lo)).sum()) / 1024.0 for lo, hi in zip(edges[:-1], edges[1:])]
starts = np.sort(sp[:, 0])
print(json.dumps({"kernel": "k_slab_sort", "config": a.config, "waves": int(sp.shape[0]), "first_start_to_last_end_us": round(float(end), 2),
                  "wave_life_us": {"mean": round(float(life.mean()), 2), "p50": round(float(np.median(life)), 2), "p99": round(float(np.percentile(life, 99)), 2), "max": round(float(life.max()), 2)},
                  "start_time_percentiles_us": {p: round(float(np.percentile(starts, p)), 2) for p in (10, 50, 90, 100)},
                  "waves_per_simd_in_20_time_bins": [round(x, 2) for x in occ]}))
