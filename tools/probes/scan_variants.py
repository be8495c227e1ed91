#!/usr/bin/env python
"""One timing point of the point-in-tet forward at a BASELINE configuration: query sort + traversal + finalize on the
workload's first input set; the traversal kernel is timed with the library's own HIP events on the launch stream, the
whole forward and the backward with HIP events around `reps` calls, and `cond` is compared with the brute-force kernel
when --check is given.

    python tools/probes/scan_variants.py [--config 2] [--reps 10] [--algo 0] [--kernel NAME] [--check]

Grid tunables are read ONCE per process (DEFTET_PIT_YZFINE / _XFINE / _GDIV / _QDIV), so a sweep is a shell loop over
processes; DEFTET_HIP_LIB=<path> times another build of the library (tools/probes/build_variant.sh); --algo 3 / 4 are the
round-3 / round-4 traversal kernels of the product library (A/B inside one build).
Prints one JSON line."""
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


def run_steplike(wl, lib, algo, reps, kernel):
    """The benchmark's regime: rotate over the input sets and run forward + backward every iteration, so nothing of a
    step is left in the 256 MiB Infinity Cache by the previous one.  Returns the traversal's average (library events) and
    the average step time."""
    def step(i):
        d = wl.sets[i % len(wl.sets)]
        outs = hip_ops.point_in_tet(d["tet"], d["pts"], want_bary=True, pred_bxt=d["pred"], want_hits=True, algo=algo, order=wl.order)
        hip_ops.point_in_tet_bwd(d["tet"], d["pts"], outs[0], d["gw"], grad_occ=d["gout"], hits=outs[3])
        return outs
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    lib.deftet_profile_select(kernel.encode())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        step(3 + i)
    b.record()
    torch.cuda.synchronize()
    tot, cnt = ctypes.c_double(0), ctypes.c_longlong(0)
    lib.deftet_profile_read(ctypes.byref(tot), ctypes.byref(cnt))
    lib.deftet_profile_select(b"")
    return tot.value / max(cnt.value, 1) * 1e3, a.elapsed_time(b) / reps * 1e3


def run(wl, lib, algo, reps, kernel):
    d = wl.sets[0]
    outs = None
    for _ in range(2):
        outs = hip_ops.point_in_tet(d["tet"], d["pts"], want_bary=True, pred_bxt=d["pred"], want_hits=True, algo=algo, order=wl.order)
    torch.cuda.synchronize()
    lib.deftet_profile_select(kernel.encode())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        outs = hip_ops.point_in_tet(d["tet"], d["pts"], want_bary=True, pred_bxt=d["pred"], want_hits=True, algo=algo, order=wl.order)
    b.record()
    torch.cuda.synchronize()
    tot, cnt = ctypes.c_double(0), ctypes.c_longlong(0)
    lib.deftet_profile_read(ctypes.byref(tot), ctypes.byref(cnt))
    lib.deftet_profile_select(b"")
    stats = hip_ops.point_in_tet_stats(wl.B, wl.T, wl.Q, algo, d["tet"].device).sum(0).tolist()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        g = hip_ops.point_in_tet_bwd(d["tet"], d["pts"], outs[0], d["gw"], grad_occ=d["gout"], hits=outs[3])
    t1.record()
    torch.cuda.synchronize()
    return outs, g, tot.value / max(cnt.value, 1) * 1e3, a.elapsed_time(b) / reps * 1e3, t0.elapsed_time(t1) / reps * 1e3, stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--algo", type=int, default=0)
    ap.add_argument("--kernel", default=None, help="traversal kernel name to time (default: the one `algo` launches in the product library)")
    ap.add_argument("--order", default=None, help="xfast: enumerate the Kuhn cubes with x fastest instead of z fastest")
    ap.add_argument("--mesh", default=None, help="cube40: the shipped QuarTet grid (res 40 sizes) instead of the Kuhn grid; shuffled: the Kuhn grid's tet list in random order")
    ap.add_argument("--tet-order", default="auto", choices=["auto", "native", "sorted", "identity", "ideal"], help="traversal order handed to the operator")
    ap.add_argument("--check", action="store_true", help="compare cond with the brute-force kernel (slow at configs[2..3])")
    ap.add_argument("--sets", type=int, default=3, help="input sets of the step-like timing (1 = everything stays in the Infinity Cache)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    cfg = dict(bench.CONFIGS[a.config], sets=max(1, a.sets))
    if a.mesh:
        cfg["mesh"] = a.mesh
    if a.order:
        cfg["order"] = a.order
    cfg["tet_order"] = a.tet_order
    wl = bench.PitWorkload(cfg, 0, dev, 1, None, pipeline=False)
    kernel = a.kernel or hip_ops.pit_kernel_name(a.algo, wl.T, wl.Q)
    outs, g, k_us, fwd_us, bwd_us, stats = run(wl, lib, a.algo, a.reps, kernel)
    step_k_us, step_us = run_steplike(wl, lib, a.algo, a.reps, kernel)
    rec = {"config": a.config, "mesh": (a.mesh or "kuhn") + ("/" + a.order if a.order else ""), "n_tet": wl.T, "algo": a.algo, "kernel": kernel,
           "tet_order": a.tet_order + ("" if a.tet_order != "auto" else (" -> native" if wl.order is None else " -> sorted")),
           "lib": os.environ.get("DEFTET_HIP_LIB", "product"),
           "env": {k: os.environ[k] for k in ("DEFTET_PIT_YZFINE", "DEFTET_PIT_XFINE", "DEFTET_PIT_GDIV", "DEFTET_PIT_QDIV") if k in os.environ},
           "traversal_us_in_step": round(step_k_us, 2), "step_us": round(step_us, 1),
           "traversal_us_warm": round(k_us, 2), "fwd_us_warm": round(fwd_us, 1), "bwd_us_warm": round(bwd_us, 1),
           "roofline_frac_of_8TBs_in_step": round(wl.dominant_bytes / (step_k_us * 1e-6) / 8e12, 4) if step_k_us > 0 else None,
           "stats_irrT_irrQ_ovf_x_x_rescanned_ovfTets": stats[:7]}
    if hasattr(lib, "deftet_point_in_tet_grid_dims"):
        rec["grid_yz_x"] = hip_ops.point_in_tet_grid(wl.T, wl.Q)
    if a.check:
        d = wl.sets[0]
        brute = hip_ops.point_in_tet(d["tet"], d["pts"], algo=1)
        rec["same_as_brute"] = bool(torch.equal(outs[0], brute))
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
