#!/usr/bin/env python
"""Per-kernel times of the query sort (k_query_bbox, k_slab_local, k_slab_sort) and of the other kernels of the step, from the
library's own HIP events around each launch, rotating input sets: python tools/probes/sort_probe.py [--config 2]"""
import argparse, ctypes, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deftet_amd import _lib, hip_ops  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--config", type=int, default=2); ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
lib = _lib.load(); dev = torch.device("cuda:0")
wl = bench.PitWorkload(dict(bench.CONFIGS[a.config]), 0, dev, 1, None, pipeline=False)
out = {}
for k in ("k_query_bbox", "k_slab_local", "k_slab_sort", hip_ops.pit_kernel_name(0, wl.T, wl.Q), "k_finalize", "k_bary_bwd_hits", "deftet::red::k_rowdot_fused"):
    for i in range(3): wl.step(i)
    torch.cuda.synchronize()
    lib.deftet_profile_select(k.encode())
    for i in range(a.reps): wl.step(3 + i)
    torch.cuda.synchronize()
    tot, cnt = ctypes.c_double(0), ctypes.c_longlong(0)
    lib.deftet_profile_read(ctypes.byref(tot), ctypes.byref(cnt)); lib.deftet_profile_select(b"")
    out[k] = round(tot.value / max(cnt.value, 1) * 1e3, 2)
out["sum_us"] = round(sum(out.values()), 1)
print(json.dumps({"config": a.config, "kernel_us_in_step": out}))
