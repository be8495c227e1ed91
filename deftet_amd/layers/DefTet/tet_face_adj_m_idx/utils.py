"""Surface-face edge adjacency with the reference's operator surface.

Mirrors /root/reference/layers/DefTet/tet_face_adj_m_idx/utils.py:37-70:
`tet_face_adj_m_f_idx(face_fx3x3) -> int64 [2, E]` (row 0 = face, row 1 = neighbour, in
row-major order of the [F,30] table), a 0-length float tensor when F == 0, backward None.
"""
import os

import torch
from torch.autograd import Function

from deftet_amd import hip_ops


class VarianceFunc(Function):
    @staticmethod
    def forward(ctx, face_fx3x3):
        n_face = face_fx3x3.shape[0]
        if n_face == 0:
            return torch.zeros(0, device=face_fx3x3.device)                       # utils.py:42-43
        n_max_nei = 30                                                            # utils.py:45
        adj_idx = hip_ops.face_edge_adj(face_fx3x3, n_max_nei)
        idx = torch.arange(0, n_face, device=face_fx3x3.device, dtype=torch.long).int()
        idx = idx.unsqueeze(-1).unsqueeze(-1).expand(-1, n_max_nei, 1)
        mask = (adj_idx >= 0)
        adj_idx = adj_idx.int().unsqueeze(-1)
        all_adj_idx = torch.cat([idx, adj_idx], dim=-1)
        all_adj_idx = all_adj_idx[mask]
        return all_adj_idx.permute(1, 0).long()                                   # utils.py:58-61

    @staticmethod
    def backward(ctx, dl_dclosest_d):
        return None


tet_face_adj_m_f_idx = VarianceFunc.apply
