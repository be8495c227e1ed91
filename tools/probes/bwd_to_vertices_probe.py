"""A1b backward onto the vertices: the fused call (deftet_point_in_tet_bwd_to_vertices_f32: compacted rows + mask, masked
vertex gather) against the two-call form (dense dL/dtet + deftet_tet_gather_bwd_f32) at BASELINE configs[2] sizes.
One JSON line per form: ms per call (HIP events over `--iters` calls), equality of the results."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deftet_amd import grids, hip_ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=70)
ap.add_argument("--queries", type=int, default=100000)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda:0")
verts, tets = grids.kuhn_grid(a.res)
pos = torch.from_numpy(grids.jittered_positions(verts, a.res, a.batch, 0.1).astype(np.float32)).to(dev)
idx = torch.from_numpy(tets.astype(np.int64)).to(dev)
q = torch.from_numpy(grids.random_queries(a.batch, a.queries)).to(dev)
V, T = pos.shape[1], idx.shape[0]
t = hip_ops.tet_gather(pos, idx)
csr = hip_ops.tet_vertex_csr(idx, V)
gen = torch.Generator(device=dev).manual_seed(3)
pred = torch.rand(a.batch, T, device=dev, generator=gen)
cond, w, occ, hits = hip_ops.point_in_tet(t, q, want_bary=True, pred_bxt=pred, want_hits=True)
gw = torch.randn(a.batch, a.queries, 4, device=dev, generator=gen)
go = torch.randn(a.batch, a.queries, device=dev, generator=gen)


def two_call():
    g_tet, _, g_pred = hip_ops.point_in_tet_bwd(t, q, cond, gw, grad_occ=go, hits=hits)
    return hip_ops.tet_gather_bwd(g_tet, csr, V), g_pred


def fused():
    g_pos, _, g_pred = hip_ops.point_in_tet_bwd_to_vertices(t, q, cond, gw, csr, V, grad_occ=go, hits=hits)
    return g_pos, g_pred


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters


ra, rb = two_call(), fused()
same = bool((ra[0] == rb[0]).all()) and bool((ra[1] == rb[1]).all())
for name, fn in (("two_call", two_call), ("fused", fused), ("two_call", two_call), ("fused", fused)):
    print(json.dumps({"form": name, "ms_per_call": round(timed(fn), 4), "res": a.res, "n_tet": T, "n_vertex": V, "n_query": a.queries,
                      "batch": a.batch, "identical_results": same, "rows_with_hits_frac": round(float((hits[:2 * a.batch * T].view(-1, 2)[:, 0] >= 0).float().mean()), 4)}),
          flush=True)
