#!/usr/bin/env python
"""Work counters of k_tri_query_coop (A9) on the geometry-step workload: needs a probe build with -DTRI_STATS,
    python -m deftet_amd.build --out tools/probes/bin/libdeftet_tristats.so -DTRI_STATS
    DEFTET_HIP_LIB=tools/probes/bin/libdeftet_tristats.so python tools/probes/tri_stats_probe.py
Prints one JSON line (per forward call over 8 shapes)."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deftet_amd import _lib  # noqa: E402

NAMES = ["wave_chunks", "rounds_64", "entries_loaded", "faces_broadcast", "drains", "first_evals_lanes", "later_eval_rounds",
         "later_evals_lanes", "wide_broadcast", "-", "live_lanes", "x_span_cells", "shell1", "shell2", "-", "-"]


def main():
    lib = _lib.load()
    fn = lib.deftet_debug_tri_stats               # only in -DTRI_STATS builds
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    wl = bench.make_workload(5, 0, torch.device("cuda:0"), 1, None)
    wl.step(0)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    fn(buf, 1)
    wl.step(1)
    torch.cuda.synchronize()
    fn(buf, 1)
    print(json.dumps({n: int(v) for n, v in zip(NAMES, buf) if n != "-"}))


if __name__ == "__main__":
    main()
