#!/usr/bin/env python
"""A/B of the point-in-tet traversal kernels at a BASELINE configuration: for every (algo, grid parameters)
combination the forward (query sort + traversal + finalize) is run on the same inputs, the traversal kernel is
timed with the library's own HIP events on the launch stream, and `cond` is compared with the default path.

    python tools/probes/scan_variants.py [--config 2] [--reps 10] [--sweep]
Prints one JSON line per variant."""
import argparse
import ctypes
import itertools
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deftet_amd import _lib, hip_ops  # noqa: E402


def set_env(**kw):
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)


def run(wl, lib, algo, reps):
    d = wl.sets[0]
    name = hip_ops.pit_kernel_name(algo).encode()
    outs = None
    for _ in range(2):
        outs = hip_ops.point_in_tet(d["tet"], d["pts"], want_bary=True, pred_bxt=d["pred"], want_hits=True, algo=algo)
    torch.cuda.synchronize()
    lib.deftet_profile_select(name)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        outs = hip_ops.point_in_tet(d["tet"], d["pts"], want_bary=True, pred_bxt=d["pred"], want_hits=True, algo=algo)
    b.record()
    torch.cuda.synchronize()
    tot, cnt = ctypes.c_double(0), ctypes.c_longlong(0)
    lib.deftet_profile_read(ctypes.byref(tot), ctypes.byref(cnt))
    lib.deftet_profile_select(b"")
    run.stats = hip_ops.point_in_tet_stats(wl.B, wl.T, wl.Q, algo, d["tet"].device).sum(0).tolist()
    # backward from this variant's hit records
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        g = hip_ops.point_in_tet_bwd(d["tet"], d["pts"], outs[0], d["gw"], grad_occ=d["gout"], hits=outs[3])
    t1.record()
    torch.cuda.synchronize()
    return outs, g, tot.value / max(cnt.value, 1) * 1e3, a.elapsed_time(b) / reps * 1e3, t0.elapsed_time(t1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--order", default=None, help="xfast: enumerate the Kuhn cubes with x fastest (the grid's run axis) instead of z fastest")
    ap.add_argument("--mesh", default=None, help="cube40: the shipped QuarTet grid (res 40 sizes) instead of the Kuhn grid")
    ap.add_argument("--algos", default="0,11,6,9,10")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    cfg = dict(bench.CONFIGS[a.config], sets=1)
    if a.mesh:
        cfg["mesh"] = a.mesh
    if a.order:
        cfg["order"] = a.order
    wl = bench.PitWorkload(cfg, 0, dev, 1, None, pipeline=False)
    set_env(DEFTET_PIT_XFINE=None, DEFTET_PIT_GDIV=None, DEFTET_PIT_QDIV=None)
    ref, gref, *_ = run(wl, lib, 0, 2)
    algos = [int(x) for x in a.algos.split(",")]
    grids = [(None, None, None)]
    if a.sweep:
        grids += [(xf, gd, qd) for xf, (gd, qd) in itertools.product((2, 4, 8), ((6, 2), (3, 1), (1.5, 0.5), (0.75, 0.25), (12, 4)))]
    for (xf, gd, qd), algo in itertools.product(grids, algos):
        set_env(DEFTET_PIT_XFINE=xf, DEFTET_PIT_GDIV=gd, DEFTET_PIT_QDIV=qd)
        try:
            outs, g, k_us, fwd_us, bwd_us = run(wl, lib, algo, a.reps)
        except Exception as e:                                   # e.g. workspace limits of a grid setting
            print(json.dumps({"algo": algo, "xfine": xf, "gdiv": gd, "qdiv": qd, "error": str(e)[:200]}), flush=True)
            continue
        same = bool(torch.equal(outs[0], ref[0]) and torch.equal(outs[1], ref[1]) and torch.equal(outs[2], ref[2]))
        gerr = float((g[0] - gref[0]).abs().max() / gref[0].abs().max())
        algo_bytes = wl.dominant_bytes
        print(json.dumps({"config": a.config, "mesh": (a.mesh or "kuhn") + ("/" + a.order if a.order else ""), "n_tet": wl.T, "algo": algo, "kernel": hip_ops.pit_kernel_name(algo), "xfine": xf, "gdiv": gd, "qdiv": qd,
                          "traversal_us": round(k_us, 2), "fwd_us": round(fwd_us, 1), "bwd_us": round(bwd_us, 1),
                          "roofline_frac_of_8TBs": round(algo_bytes / (k_us * 1e-6) / 8e12, 4), "same_as_default": same, "stats_irrT_irrQ_ovf_defer_grpRescan_tetRescan": run.stats[:6],
                          "bwd_rel_diff": gerr}), flush=True)
    set_env(DEFTET_PIT_XFINE=None, DEFTET_PIT_GDIV=None, DEFTET_PIT_QDIV=None)


if __name__ == "__main__":
    main()
