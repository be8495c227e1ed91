"""Host-side cost of the plumbing around one library call (us per call, CPU wall time, GPU idle): what a launch costs the
Python thread before the kernel is even enqueued.   python tools/probes/host_call_probe.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from deftet_amd import _lib, hip_ops

dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)
N = 2000


def per_call(fn):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return round(dt / N * 1e6, 2)


def ctx():
    with torch.cuda.device(dev):
        pass


out = {
    "with torch.cuda.device(dev)": per_call(ctx),
    "torch.cuda.current_stream(dev).cuda_stream": per_call(lambda: torch.cuda.current_stream(dev).cuda_stream),
    "_lib.current_stream(dev)": per_call(lambda: _lib.current_stream(dev)),
    "_lib.ptr(x)": per_call(lambda: _lib.ptr(x)),
    "_lib.workspace(dev, 1024)": per_call(lambda: _lib.workspace(dev, 1024)),
    "_lib.require_gpu(x, x, x)": per_call(lambda: _lib.require_gpu(x, x, x)),
    "torch.empty(64)": per_call(lambda: torch.empty(64, device=dev)),
    "torch.zeros(64) (one launch)": per_call(lambda: torch.zeros(64, device=dev)),
    "x + x (one launch)": per_call(lambda: x + x),
}
pos = torch.zeros(1, 8, 3, device=dev)
idx = torch.zeros(1, 4, dtype=torch.int64, device=dev)
out["hip_ops.tet_gather (tiny: one library launch)"] = per_call(lambda: hip_ops.tet_gather(pos, idx))
w = torch.zeros(2, 16, device=dev)
out["hip_ops.rowdot (tiny)"] = per_call(lambda: hip_ops.rowdot(w, w, w, w))
print(json.dumps(out))
