#!/usr/bin/env python
"""Per-kernel time of the rasterizer at BASELINE configs[4] by the library's own events (deftet_profile_select): forward kernels
for one saturation policy, then the backward's.  DEFTET_HIP_LIB=<probe build> times another library.  One JSON line."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deftet_amd import _lib  # noqa: E402
from deftet_amd.render import deftet_sparse_render  # noqa: E402
from tests.test_raster_gpu import pixel_grid, projected_grid  # noqa: E402


def main():
    policy = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    reps = 8
    lib = _lib.load()
    dev = torch.device("cuda:0")
    fz, fxy, ff = projected_grid(70)
    pix, rngs = pixel_grid(512)
    t = [torch.from_numpy(x).to(dev) for x in (pix, rngs, fz, fxy, ff)]
    t[3].requires_grad_(True)
    t[4].requires_grad_(True)
    feat, face = deftet_sparse_render(*t, knum=64, policy=policy)
    go = torch.rand_like(feat)

    def step():
        f, _ = deftet_sparse_render(*t, knum=64, policy=policy)
        torch.autograd.grad(f, (t[3], t[4]), go)

    step()
    torch.cuda.synchronize()
    out = {"lib": os.path.basename(os.environ.get("DEFTET_HIP_LIB", "product")), "policy": ["nearest", "first"][policy]}
    for k in ("k_pix_raster", "k_pix_emit", "k_bwd_sorted", "k_bwd_runs"):      # (the backward runs ONE of the last two: the other reads 0)
        lib.deftet_profile_select(k.encode())
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        tot, cnt = ctypes.c_double(0), ctypes.c_longlong(0)
        lib.deftet_profile_read(ctypes.byref(tot), ctypes.byref(cnt))
        out[k + "_us"] = round(tot.value / max(cnt.value, 1) * 1e3, 1)
    lib.deftet_profile_select(b"")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        step()
    e1.record()
    torch.cuda.synchronize()
    out["step_ms"] = round(e0.elapsed_time(e1) / reps, 3)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
