#!/usr/bin/env python
"""Work counters of k_pix_raster at BASELINE configs[4] for both saturation policies; needs a probe build:
    python -m deftet_amd.build --out tools/probes/bin/libdeftet_raststats.so -DRAST_STATS
    DEFTET_HIP_LIB=tools/probes/bin/libdeftet_raststats.so python tools/probes/rast_stats_probe.py"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from deftet_amd import _lib  # noqa: E402
from deftet_amd.render import deftet_sparse_render  # noqa: E402
from tests.test_raster_gpu import pixel_grid, projected_grid  # noqa: E402

NAMES = ["wave_chunks", "batches", "faces_broadcast", "live_lanes", "wide_faces_tested", "records_replaced", "max_faces_broadcast_by_a_wave", "longest_wave_cycles"]


def main():
    lib = _lib.load()
    fn = lib.deftet_debug_rast_stats
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    dev = torch.device("cuda:0")
    fz, fxy, ff = projected_grid(70)
    pix, rngs = pixel_grid(512)
    t = [torch.from_numpy(x).to(dev) for x in (pix, rngs, fz, fxy, ff)]
    buf = (ctypes.c_ulonglong * 8)()
    for policy in (0, 1):
        deftet_sparse_render(*t, knum=64, policy=policy)
        torch.cuda.synchronize()
        fn(buf, 1)
        _, face = deftet_sparse_render(*t, knum=64, policy=policy)
        torch.cuda.synchronize()
        fn(buf, 1)
        rec = {n: int(v) for n, v in zip(NAMES, buf) if n != "-"}
        rec["policy"] = ["nearest", "first"][policy]
        rec["hits_recorded"] = int((face >= 0).sum().item())
        print(json.dumps(rec))
    # how many faces cover a pixel at all (knum large, a sample of the pixels)
    sub = [x[:, ::37].contiguous() for x in t[:2]]
    _, face = deftet_sparse_render(sub[0], sub[1], *t[2:], knum=1200)
    n = (face >= 0).sum(-1).float()
    print(json.dumps({"covering_faces_per_pixel": {"mean": n.mean().item(), "max": n.max().item(), "p50": n.median().item(),
                                                   "frac_over_64": (n > 64).float().mean().item()}}))


if __name__ == "__main__":
    main()
