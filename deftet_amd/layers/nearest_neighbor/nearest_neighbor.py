"""1-NN index behind the reference's `layers.nearest_neighbor` surface.

Replaces layers/nearest_neighbor/nearest_neighbor.py:21-60 of the reference: `NearestNeighbor()(q, pts)` with
q float [B, N, 3] and pts float [B, M, 3] gives int64 [B, N], the index of the nearest point of the same batch entry
(lowest index on ties, as the reference kernel's strict `<` scan does).  The search itself is
`deftet_nn_index_f32` (exact grid search, `deftet_amd/csrc/surface_ops.hip`).  Like the reference, the operator has no
gradient: asking for one raises NotImplementedError; malformed shapes fail an assertion.
"""
import torch

from deftet_amd import hip_ops


def _check_shapes(q, pts):
    assert q.dim() == 3 and pts.dim() == 3, "expected [B, N, 3] queries and [B, M, 3] points"
    assert q.shape[2] == 3, "Currently only 3D points are supported"
    assert pts.shape[0] == q.shape[0], "queries and points differ in batch size"
    assert pts.shape[2] == q.shape[2], "queries and points differ in dimension"


class NearestNeighborFunction(torch.autograd.Function):
    """`apply(queries, points)` — the name and call the reference's callers use."""

    @staticmethod
    def forward(ctx, queries, points):
        _check_shapes(queries, points)
        idx = hip_ops.nn_index(queries, points)          # int32 [B, N]
        return idx.to(torch.int64)

    @staticmethod
    def backward(*grads):
        raise NotImplementedError


class NearestNeighbor(torch.nn.Module):
    def forward(self, queries, points):
        # [B, N, 3], [B, M, 3] -> [B, N] (int64)
        return NearestNeighborFunction.apply(queries, points)
