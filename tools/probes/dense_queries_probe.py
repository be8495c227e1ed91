"""Many queries per tet (the records overflow, the backward's list rescan takes over): forward / backward times with and without the
forward's hit records.  Since round 5 the library itself takes the per-tet lists when n_query > 2 n_tet, records or not: before
that rule the first column read 111 / 51 / 204 ms at 16.7 / 77 / 67 queries per tet.  python tools/probes/dense_queries_probe.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deftet_amd import grids, hip_ops
dev = torch.device("cuda:0")
import argparse
ap = argparse.ArgumentParser(); ap.add_argument('--sweep', action='store_true'); a = ap.parse_args()
cases = ((70, 100000, 2), (40, 100000, 2), (30, 60000, 2), (20, 100000, 2), (12, 100000, 2), (20, 400000, 1))
if a.sweep:
    cases = tuple((40, q, b) for b in (8, 2) for q in (25000, 50000, 75000, 100000, 125000, 144000)) + ((70, 100000, 8), (70, 300000, 8), (70, 500000, 8))
for res, Q, B in cases:
    tet, pts, _, _ = grids.make_case(res, Q, B, 0.1)
    t, p = torch.from_numpy(tet).to(dev), torch.from_numpy(pts).to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    gw, go = torch.randn(B, Q, 4, device=dev, generator=g), torch.randn(B, Q, device=dev, generator=g)
    pred = torch.rand(B, t.shape[1], device=dev, generator=g)
    def timeit(f, n=5):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    out = {}
    def fwd(): out["f"] = hip_ops.point_in_tet(t, p, want_bary=True, pred_bxt=pred, want_hits=True)
    tf = timeit(fwd)
    cond, w, occ, hits = out["f"]
    tb_hits = timeit(lambda: hip_ops.point_in_tet_bwd(t, p, cond, gw, grad_occ=go, hits=hits))
    tb_list = timeit(lambda: hip_ops.point_in_tet_bwd(t, p, cond, gw, grad_occ=go))
    st = hip_ops.point_in_tet_stats(B, t.shape[1], Q, 0, dev)
    a, b2 = hip_ops.point_in_tet_bwd(t, p, cond, gw, grad_occ=go, hits=hits), hip_ops.point_in_tet_bwd(t, p, cond, gw, grad_occ=go)
    err = (a[0] - b2[0]).abs().max().item() / b2[0].abs().max().item()
    print("res %d T %d Q %d B %d (%.1f queries per tet): fwd %.3f ms, bwd with hit records %.3f ms, bwd without %.3f ms; %s; records vs lists %.1e"
          % (res, t.shape[1], Q, B, Q / t.shape[1], tf, tb_hits, tb_list, "lists either way" if Q > 2 * t.shape[1] else "records used", err), flush=True)
