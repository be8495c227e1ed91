#!/usr/bin/env python
"""The geometry side of one DefTet training step (layers/DefTet/deftet.py:52-130,
forward_surface_align without the networks), end to end on the HIP operators:

    vertices --gather--> tets --check_sign--> centroid occupancy --get_boundary_index--> surface faces
             tets + query points --point_in_tet_occ--> (index, weights, pasted occupancy)
             tets --energies--> (volume variance, AMIPS, edge length)
    loss.backward(): d/d tets of all of it --gather_bwd--> d/d vertices (no atomics anywhere)

Prints one JSON line with the time of each stage at BASELINE configs[2] sizes.
    python tools/step_demo.py [--res 70 --batch 8 --queries 100000] [--surface [--gt-points 100000]]

--surface times the WHOLE geometry side of a training step: `DefTet.forward_surface_align` (gather, check_sign,
boundary faces, the three energies AND the per-shape surface terms A8 face adjacency / A9 point-to-surface / A10
nearest neighbour against `--gt-points` ground-truth surface points per shape) plus the occupancy query
(point-in-tet + weights + paste_occ), and the backward of all of it down to the vertices."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deftet_amd import grids, hip_ops  # noqa: E402
from deftet_amd.layers.DefTet.deftet import DefTet  # noqa: E402
from deftet_amd.layers.DefTet.check_condition_tetrahedron_base.utils import point_in_tet_occ  # noqa: E402


def build_case(res, B, Q, dev):
    verts, tets = grids.kuhn_grid(res)
    pos = torch.from_numpy(grids.jittered_positions(verts, res, B).astype(np.float32)).to(dev)
    idx = torch.from_numpy(tets.astype(np.int64)).to(dev)
    f3, t2 = hip_ops.tet_to_face(idx.int(), verts.shape[0], dev)[:2]
    # ground-truth surface: boundary of the tets of the UNJITTERED grid inside a sphere (a closed mesh)
    occ0 = torch.from_numpy(np.linalg.norm((verts - 0.5)[tets].mean(1), axis=1) < 0.3).to(dev)
    gt_faces = f3[occ0[t2].sum(1) == 1].contiguous()
    gt_verts = torch.from_numpy((verts - 0.5).astype(np.float32)).to(dev)
    pts = torch.from_numpy(grids.random_queries(B, Q)).to(dev)
    rest = gt_verts[idx] * 20
    # (.contiguous(): torch.inverse hands back column-major matrices, which every energies call would copy again)
    inv_v = torch.inverse(torch.stack([rest[:, 1] - rest[:, 0], rest[:, 2] - rest[:, 0], rest[:, 3] - rest[:, 0]], 1)).contiguous()
    return pos, idx, f3, t2, gt_verts, gt_faces, pts, inv_v


def run_step(m, pos, idx, f3, t2, gt_verts, gt_faces, pts, inv_v, pred, times=None):
    def mark(name, t0):
        if times is not None:
            torch.cuda.synchronize()
            times[name] = times.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()

    B = pos.shape[0]
    t0 = time.perf_counter()
    tet = m.gather_tet_pos(pos, idx[None].expand(B, -1, -1) if idx.dim() == 2 else idx)
    t0 = mark("gather", t0)
    center_occ = m.check_tet_inside_sdfs(tet.detach(), ([gt_verts[None]] * B, [[gt_faces]] * B))
    t0 = mark("check_sign", t0)
    boundary = m.get_boundary_index(f3, t2, center_occ.squeeze(-1))
    t0 = mark("boundary_index", t0)
    cond, w, occ = point_in_tet_occ(tet, pts, pred)
    t0 = mark("point_in_tet_occ", t0)
    vv, am, el = m.energies(tet, inv_v)
    t0 = mark("energies", t0)
    loss = (w * w).sum() + (occ - 0.5).pow(2).sum() + 1e-3 * am.sum() + 1e-3 * el.sum() + 1e-6 * vv.sum()
    t0 = mark("loss (torch)", t0)
    loss.backward()
    mark("backward (all ops + gather_bwd)", t0)
    return loss, boundary, cond


FUSED_QUERY_BWD = os.environ.get("DEFTET_STEP_FUSED_BWD", "1") not in ("", "0")


def run_full_step(m, pos, idxB, f3, t2, gt_verts, gt_faces, pts, inv_v, pred, gt_pts):
    """forward_surface_align (training branch) + occupancy query + backward."""
    B = pos.shape[0]
    m.inverse_v = inv_v
    tet = m.gather_tet_pos(pos, idxB)                                  # once: the module and the occupancy query share it
    out = m.forward_surface_align(pos, pts, tetrahedron_bxfx4=idxB, mesh_list=([gt_verts[None]] * B, [[gt_faces]] * B),
                                  gt_surface_points=gt_pts, tet_face_bxfx3=f3[None].expand(B, -1, -1),
                                  tet_face_tet_bx4fx2=t2[None].expand(B, -1, -1), tet_bxfx4x3=tet)
    amips, edge, vvar, analytic, normal, center_occ, boundary, chamfer, _ = out
    if FUSED_QUERY_BWD:     # the query's gradient goes straight to the vertices (no dense per-tet gradient in between)
        cond, w, occ = m.occupancy_query(pos, idxB, pts, pred, tet_bxfx4x3=tet)
    else:                   # DEFTET_STEP_FUSED_BWD=0: the two-stage chain (dense dL/dtet, summed with the energies' by autograd, then the gather's backward)
        cond, w, occ = point_in_tet_occ(tet, pts, pred)
    loss = standin_loss(w, occ, amips, edge, vvar, chamfer, analytic, normal)
    loss.backward()
    return loss, boundary


class _SquareSums(torch.autograd.Function):
    """[R] = sum(w[r]^2) + sum((occ[r] - 0.5)^2): the two big reductions of the stand-in loss as ONE fused row dot of the
    library (+ one elementwise launch), two launches back."""

    @staticmethod
    def forward(ctx, w, occ):
        from deftet_amd import hip_ops
        d = occ - 0.5
        ctx.save_for_backward(w, d)
        return hip_ops.rowdot(w, w, d, d)

    @staticmethod
    def backward(ctx, g):
        w, d = ctx.saved_tensors
        g2 = g + g
        return g2.reshape(-1, *([1] * (w.dim() - 1))) * w, g2.reshape(-1, *([1] * (d.dim() - 1))) * d


_COEF = {}


def standin_loss(w, occ, amips, edge, vvar, chamfer, analytic, normal):
    """sum(w^2) + sum((occ - 0.5)^2) + 1e-3 sum(amips) + 1e-3 sum(edge) + 1e-6 sum(vvar) + sum(chamfer) + sum(analytic) +
    sum(normal) — what a training loop does with the operators' outputs, written as four launches (fused row dot,
    concatenation of the scalar terms, one dot product with a constant weight vector) instead of the 23 elementwise /
    reduction launches (and ~30 in its backward) the literal expression costs."""
    parts = [_SquareSums.apply(w, occ)] + [t.reshape(-1) for t in (amips, edge, vvar, chamfer, analytic, normal)]
    key = (w.device, tuple(p.numel() for p in parts))
    coef = _COEF.get(key)
    if coef is None:
        weights = (1.0, 1e-3, 1e-3, 1e-6, 1.0, 1.0, 1.0)
        coef = _COEF[key] = torch.cat([torch.full((p.numel(),), c, device=w.device) for p, c in zip(parts, weights)])
    return torch.dot(torch.cat(parts), coef)


def surface_main(a, dev):
    from deftet_amd import surface_losses
    case = build_case(a.res, a.batch, a.queries, dev)
    pos0, idx, f3, t2, gt_verts, gt_faces, pts, inv_v = case
    B = a.batch
    per_face = max(1, a.gt_points // max(1, gt_faces.shape[0]))
    tri = gt_verts[gt_faces.long()][None].expand(B, -1, -1, -1)
    gt_pts = surface_losses.sample_on_faces(tri, per_face, torch.Generator(device=dev).manual_seed(5)).reshape(B, -1, 3).contiguous()
    pos = pos0.clone().requires_grad_(True)
    pred = torch.rand(B, idx.shape[0], device=dev, requires_grad=True)
    m = DefTet(device=dev)
    idxB = idx[None].expand(B, -1, -1).contiguous()
    for _ in range(2):
        pos.grad = None
        pred.grad = None
        _, boundary = run_full_step(m, pos, idxB, f3, t2, gt_verts, gt_faces, pts, inv_v, pred, gt_pts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        pos.grad = None
        pred.grad = None
        run_full_step(m, pos, idxB, f3, t2, gt_verts, gt_faces, pts, inv_v, pred, gt_pts)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"op": "DefTet geometry step incl. surface terms (forward_surface_align + point-in-tet occupancy, fwd + bwd to vertices)",
                      "res": a.res, "batch": B, "n_tet": int(idx.shape[0]), "n_query": a.queries, "n_gt_points": int(gt_pts.shape[1]),
                      "n_gt_face": int(gt_faces.shape[0]), "n_boundary_face_per_shape": [int(x.shape[0]) for x in boundary],
                      "surface_streams": min(B, 8), "ms_per_step": round(total * 1e3, 3)}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=70)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--queries", type=int, default=100000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--surface", action="store_true")
    ap.add_argument("--gt-points", type=int, default=100000)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.surface:
        return surface_main(a, dev)
    case = build_case(a.res, a.batch, a.queries, dev)
    pos = case[0].clone().requires_grad_(True)
    pred = torch.rand(a.batch, case[1].shape[0], device=dev, requires_grad=True)
    m = DefTet(device=dev)
    idxB = case[1][None].expand(a.batch, -1, -1).contiguous()
    for _ in range(2):
        pos.grad = None
        run_step(m, pos, idxB, *case[2:], pred)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        pos.grad = None
        pred.grad = None
        run_step(m, pos, idxB, *case[2:], pred)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / a.steps
    times = {}
    for _ in range(a.steps):
        pos.grad = None
        pred.grad = None
        run_step(m, pos, idxB, *case[2:], pred, times)
    print(json.dumps({"op": "DefTet geometry step (gather, check_sign, boundary, point-in-tet+paste, energies, backward to vertices)",
                      "res": a.res, "batch": a.batch, "n_tet": int(case[1].shape[0]), "n_query": a.queries,
                      "n_gt_face": int(case[5].shape[0]), "ms_per_step": round(total * 1e3, 3),
                      "stage_ms_with_sync": {k: round(v / a.steps * 1e3, 3) for k, v in times.items()}}))


if __name__ == "__main__":
    main()
