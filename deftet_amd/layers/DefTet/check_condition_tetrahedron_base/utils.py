"""Point-in-tet occupancy query with the reference's operator surface.

Mirrors /root/reference/layers/DefTet/check_condition_tetrahedron_base/utils.py:38-62:
`check_condition_f_base(tet_bxfx4x3, point_pos_bxnx3) -> condition_bxnx1` (float32 tensor
holding the lowest containing tet index or -1), a torch.autograd.Function whose backward
returns (None, None) exactly like the reference (utils.py:55-58).

The arithmetic runs in libdeftet_hip.so (deftet_amd/csrc/point_in_tet.hip); there is no
CPU fallback.  `point_in_tet_bary` is the build-defined differentiable extension
(SURVEY.md section 8 row A1b): index + barycentric weights with a backward to the tet
vertex positions (and optionally the query points).
"""
import torch
from torch.autograd import Function

from deftet_amd import hip_ops


class TriRender2D(Function):
    @staticmethod
    def forward(ctx, tet_bxfx4x3, point_pos_bxnx3, topology=None):
        # the reference also builds an (unused) [B,T,6] bbox tensor here (utils.py:47);
        # the kernel never read it (check_condition_tet_for.cu:154-164), so it is dropped.
        # order="auto": the traversal order is decided once per grid (static topology) from how coherently the list is numbered;
        # query_box="track": the query grid spans the box the previous call measured (one launch fewer); neither ever changes
        # the result.  topology (not in the reference's signature, optional): what identifies the tet list — see hip_ops.auto_tet_order
        return hip_ops.point_in_tet(tet_bxfx4x3, point_pos_bxnx3, order="auto", query_box="track", topology=topology)

    @staticmethod
    def backward(ctx, condition_bxnx1):
        return None, None, None


def check_condition_f_base(tet_bxfx4x3, point_pos_bxnx3, topology=None):
    """condition_bxnx1 (utils.py:38-62).  `topology` is optional and build-defined: a TetTopology / hashable key / index tensor
    that identifies the tet list, so that the traversal-order decision is cached per topology instead of per size."""
    return TriRender2D.apply(tet_bxfx4x3, point_pos_bxnx3, topology)


class PointInTetBary(Function):
    """(tet [B,T,4,3], pts [B,Q,3]) -> (condition [B,Q,1], weights [B,Q,4]).

    weights follow utils/tet_utils.py:28-45 (bary_centric_tet) for the hit tet, zeros for
    misses; gradients flow to tet (atomic scatter over the hit tets) and to pts."""

    @staticmethod
    def forward(ctx, tet_bxfx4x3, point_pos_bxnx3):
        # hit records only where the backward reads them (<= 2 queries per tet; beyond, it takes per-tet lists: hip_ops.bwd_uses_records)
        rec = hip_ops.bwd_uses_records(tet_bxfx4x3.shape[1], point_pos_bxnx3.shape[1])
        out = hip_ops.point_in_tet(tet_bxfx4x3, point_pos_bxnx3, want_bary=True, want_hits=rec, order="auto", query_box="track")
        cond, w, hits = out if rec else (out + (None,))
        ctx.save_for_backward(tet_bxfx4x3, point_pos_bxnx3, cond, hits)
        ctx.mark_non_differentiable(cond)
        return cond, w

    @staticmethod
    def backward(ctx, _grad_cond, grad_w):
        tet, pts, cond, hits = ctx.saved_tensors
        need_pts = ctx.needs_input_grad[1]
        g_tet, g_pts = hip_ops.point_in_tet_bwd(tet, pts, cond, grad_w, want_grad_pts=need_pts, hits=hits)
        return (g_tet if ctx.needs_input_grad[0] else None), g_pts


point_in_tet_bary = PointInTetBary.apply


class PointInTetOcc(Function):
    """Fused A1 + A1b + paste_occ: (tet [B,T,4,3], pts [B,Q,3], pred_tet_occ [B,T]) ->
    (condition [B,Q,1], weights [B,Q,4], occ [B,Q]).  One forward (the paste gather rides in
    the finalize kernel) and one backward (grad_tet and grad_pred come out of the same per-tet
    gather pass, no floating-point atomics).  `condition` keeps its -1 entries."""

    @staticmethod
    def forward(ctx, tet_bxfx4x3, point_pos_bxnx3, pred_tet_occ):
        rec = hip_ops.bwd_uses_records(tet_bxfx4x3.shape[1], point_pos_bxnx3.shape[1])
        out = hip_ops.point_in_tet(tet_bxfx4x3, point_pos_bxnx3, want_bary=True, pred_bxt=pred_tet_occ, want_hits=rec, order="auto", query_box="track")
        cond, w, occ, hits = out if rec else (out + (None,))
        ctx.save_for_backward(tet_bxfx4x3, point_pos_bxnx3, cond, hits)
        ctx.mark_non_differentiable(cond)
        return cond, w, occ

    @staticmethod
    def backward(ctx, _grad_cond, grad_w, grad_occ):
        tet, pts, cond, hits = ctx.saved_tensors
        g_tet, g_pts, g_pred = hip_ops.point_in_tet_bwd(tet, pts, cond, grad_w, want_grad_pts=ctx.needs_input_grad[1],
                                                        grad_occ=grad_occ, hits=hits)
        return (g_tet if ctx.needs_input_grad[0] else None), g_pts, (g_pred if ctx.needs_input_grad[2] else None)


point_in_tet_occ = PointInTetOcc.apply


class PointInTetOccVertices(Function):
    """PointInTetOcc for a caller that owns the vertex -> tet gather as well (layers/DefTet/deftet.py:65-68 followed by the query):
    (vertice_pos [B,V,3], pts [B,Q,3], pred_tet_occ [B,T], topology[, tet_bxfx4x3]) -> (condition, weights, occ).

    `topology` is a deftet_amd.layers.DefTet.deftet.TetTopology (index list + incidence CSR).  tet_bxfx4x3: the positions gathered
    from `vertice_pos` with that topology, when the caller has them already (their VALUES are used; the gradient goes to
    vertice_pos).  The backward never writes the dense [B,T,4,3] gradient: deftet_point_in_tet_bwd_to_vertices_f32 keeps the rows
    of the tets that accepted a query and sums them per vertex in the CSR's order — the same bits as PointInTetOcc's backward
    followed by the gather's."""

    @staticmethod
    def forward(ctx, vertice_pos, point_pos_bxnx3, pred_tet_occ, topology, tet_bxfx4x3=None):
        tet = tet_bxfx4x3.detach() if tet_bxfx4x3 is not None else hip_ops.tet_gather(vertice_pos, topology.tet_idx)
        rec = hip_ops.bwd_uses_records(tet.shape[1], point_pos_bxnx3.shape[1])
        out = hip_ops.point_in_tet(tet, point_pos_bxnx3, want_bary=True, pred_bxt=pred_tet_occ, want_hits=rec, order="auto",
                                   query_box="track", topology=topology)
        cond, w, occ, hits = out if rec else (out + (None,))
        ctx.save_for_backward(tet, point_pos_bxnx3, cond, hits)
        ctx.topology = topology
        ctx.n_vertex = vertice_pos.shape[1]
        ctx.mark_non_differentiable(cond)
        return cond, w, occ

    @staticmethod
    def backward(ctx, _grad_cond, grad_w, grad_occ):
        tet, pts, cond, hits = ctx.saved_tensors
        g_pos, g_pts, g_pred = hip_ops.point_in_tet_bwd_to_vertices(tet, pts, cond, grad_w, ctx.topology.csr, ctx.n_vertex,
                                                                    want_grad_pts=ctx.needs_input_grad[1], grad_occ=grad_occ, hits=hits)
        return (g_pos if ctx.needs_input_grad[0] else None), g_pts, (g_pred if ctx.needs_input_grad[2] else None), None, None


point_in_tet_occ_vertices = PointInTetOccVertices.apply


class PasteOcc(Function):
    """DefTet.paste_occ (layers/DefTet/deftet.py:132-136) as one fused gather with its
    scatter-add backward; `condition` is clamped in place like the reference does."""

    @staticmethod
    def forward(ctx, pred_tet_occ, condition):
        out = hip_ops.paste_occ_fwd(pred_tet_occ, condition)
        ctx.save_for_backward(condition)
        ctx.n_tet = pred_tet_occ.shape[1]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (condition,) = ctx.saved_tensors
        return hip_ops.paste_occ_bwd(condition, grad_out, ctx.n_tet), None


paste_occ = PasteOcc.apply
