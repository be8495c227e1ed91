"""Surface-face edge adjacency behind the reference's `tet_face_adj_m_f_idx`.

Replaces layers/DefTet/tet_face_adj_m_idx/utils.py:37-70 of the reference: `tet_face_adj_m_f_idx(face_fx3x3)` with
float [F, 3, 3] gives int64 [2, E] — row 0 the face, row 1 one of its edge neighbours, pairs listed in row-major order
of the [F, 30] neighbour table the kernel fills (`deftet_face_edge_adj_f32`; −1 = unused slot); for F == 0 the
reference hands back an empty FLOAT tensor, and so does this.  No gradient (backward returns None).
"""
import torch
from torch.autograd import Function

from deftet_amd import hip_ops

_SLOTS = 30                                                  # neighbour slots per face (reference utils.py:45)


class VarianceFunc(Function):
    @staticmethod
    def forward(ctx, face_fx3x3):
        if face_fx3x3.shape[0] == 0:
            return torch.zeros(0, device=face_fx3x3.device)
        table = hip_ops.face_edge_adj(face_fx3x3, _SLOTS)   # float [F, 30]
        # nonzero() walks the table row-major: the order boolean-mask indexing gave the reference
        face, slot = (table >= 0).nonzero(as_tuple=True)
        return torch.stack([face, table[face, slot].to(torch.int64)])

    @staticmethod
    def backward(ctx, _grad):
        return None


tet_face_adj_m_f_idx = VarianceFunc.apply
