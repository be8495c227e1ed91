#!/usr/bin/env python
"""A11 device time without autograd plumbing: the C-ABI forward (deftet_tet_energies_fwd_f32) and backward called
back to back at res 70, batch 8, rotating over `--sets` input sets (3 x 99 MB does not fit the 256 MiB Infinity Cache),
HIP events around `--reps` iterations.  Prints one JSON line; run it under `rocprofv3 --kernel-trace --stats` for the
per-kernel split."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deftet_amd import _lib, grids  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=70)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--sets", type=int, default=3)
    ap.add_argument("--reps", type=int, default=60)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    verts, tets = grids.kuhn_grid(a.res)
    T, B = tets.shape[0], a.batch
    sets = [torch.from_numpy(grids.gather_tets(grids.jittered_positions(verts, a.res, B, 0.1, seed0=1000 + 100 * s), tets)).to(dev) for s in range(a.sets)]
    tets_d = torch.from_numpy(tets).to(dev).long()
    rest = torch.from_numpy((verts - 0.5).astype(np.float32)).to(dev)[tets_d] * 20
    inv = torch.inverse(torch.stack([rest[:, 1] - rest[:, 0], rest[:, 2] - rest[:, 0], rest[:, 3] - rest[:, 0]], 1)).contiguous()
    out = torch.empty(B, 3, device=dev)
    stats = torch.empty(B, 8, device=dev, dtype=torch.float64)
    gout = torch.ones(B, 3, device=dev)
    grad = torch.empty_like(sets[0])
    ws = _lib.workspace(dev, lib.deftet_tet_energies_workspace_bytes2(B, T))
    st = _lib.current_stream(dev)

    def fwd(t):
        _lib.check(lib.deftet_tet_energies_fwd_f32(_lib.ptr(t), _lib.ptr(inv), _lib.ptr(out), _lib.ptr(stats), B, T, 4, 4, 20.0,
                                                   _lib.ptr(ws), ws.numel(), st), "fwd")

    def bwd(t):
        _lib.check(lib.deftet_tet_energies_bwd_f32(_lib.ptr(t), _lib.ptr(inv), _lib.ptr(stats), _lib.ptr(gout), _lib.ptr(grad), B, T, 4, 4, 20.0, st), "bwd")

    def timed(fn):
        for i in range(3):
            fn(sets[i % len(sets)])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(a.reps):
            fn(sets[i % len(sets)])
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    f, b_ = timed(fwd), timed(bwd)
    both = timed(lambda t: (fwd(t), bwd(t)))
    rec = {"op": "tet_energies (C-ABI, no autograd)", "res": a.res, "batch": B, "n_tet": T, "input_sets": a.sets,
           "fwd_ms": round(f, 4), "bwd_ms": round(b_, 4), "fwd_plus_bwd_ms": round(both, 4),
           "algorithmic_mb": {"fwd": round(B * T * 48 / 1e6, 1), "bwd": round(B * T * 96 / 1e6, 1)},
           "roofline_frac_of_8TBs": {"fwd": round(B * T * 48 / (f * 1e-3) / 8e12, 3), "bwd": round(B * T * 96 / (b_ * 1e-3) / 8e12, 3),
                                     "fwd_plus_bwd": round(B * T * 144 / (both * 1e-3) / 8e12, 3)}}
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
