"""Brute-force nearest-neighbour index with the reference's operator surface.

Mirrors /root/reference/layers/nearest_neighbor/nearest_neighbor.py:21-60:
`NearestNeighbor()(queries [B,N,3], points [B,M,3]) -> int64 [B,N]`; backward raises
NotImplementedError like the reference.
"""
import torch

from deftet_amd import hip_ops


class NearestNeighborFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, queries, points):
        batch_size, num_queries, dim = queries.shape
        _, num_points, _ = points.shape
        assert dim == 3, "Currently only 3D points are supported"
        assert batch_size == points.shape[0]
        assert dim == points.shape[2]
        return hip_ops.nn_index(queries, points).long()

    @staticmethod
    def backward(*args):
        raise NotImplementedError


class NearestNeighbor(torch.nn.Module):
    def forward(self, queries, points):
        """
        queries.shape = (batch_size, num_queries, 3)
        points.shape = (batch_size, num_points, 3)
        return shape = (batch_size, num_queries)
        """
        return NearestNeighborFunction.apply(queries, points)
