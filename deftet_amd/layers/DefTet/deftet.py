"""`DefTet` geometry module with the reference's method names and signatures
(/root/reference/layers/DefTet/deftet.py), HIP-backed for the per-tetrahedron pieces:

    get_boundary_index / get_internal_index   (:186-203)  -> deftet_boundary_index_i64
    paste_occ                                 (:132-136)  -> deftet_paste_occ_*_f32
    volume_variance / amips_energy / edge_length (:239-338) -> deftet_tet_energies_*_f32
    tet_inverse_v / my_inverse                (:205-233,:300-318)  torch (init-time, T 3x3 inverses)
    point queries                             (:110-112)  -> check_condition_f_base
    gather_tet_pos (vertex -> tet gather)     (:65-68)    -> deftet_tet_gather_{fwd,bwd}_f32

    check_tet_inside_sdfs                     (:33-49)    -> deftet_check_sign_f32 (Kaolin's
                                                             check_sign restated; parity unpinned)

`forward_surface_align` itself (network glue around these calls) is not provided; the pieces it
composes are all here or in deftet_amd.layers / deftet_amd.utils.
"""
import torch
import torch.nn as nn

from deftet_amd import hip_ops
from deftet_amd.layers.DefTet.check_condition_tetrahedron_base.utils import check_condition_f_base, paste_occ

EPS = 1e-10


class _TetGather(torch.autograd.Function):
    """tet_bxfx4x3 = gather(vertice_pos, tetrahedron) with an atomic-free, deterministic backward."""

    @staticmethod
    def forward(ctx, vertice_pos, tet_idx, csr):
        ctx.csr = csr
        ctx.n_vertex = vertice_pos.shape[1]
        return hip_ops.tet_gather(vertice_pos, tet_idx)

    @staticmethod
    def backward(ctx, grad_tet):
        return hip_ops.tet_gather_bwd(grad_tet.contiguous(), ctx.csr, ctx.n_vertex), None, None


class TetTopology:
    """Per-topology cache for the vertex<->tet gather: the incidence CSR is built once (the tet
    list is static during training, train_multigpu.py:72-77) and reused by every backward."""

    def __init__(self, tet_idx, n_vertex):
        self.tet_idx = tet_idx.long().contiguous()
        self.n_vertex = int(n_vertex)
        self.csr = hip_ops.tet_vertex_csr(self.tet_idx, self.n_vertex)

    def gather(self, vertice_pos):
        return _TetGather.apply(vertice_pos, self.tet_idx, self.csr)


class DefTet(nn.Module):
    def __init__(self, device=None):
        super(DefTet, self).__init__()
        self.pow = 4
        self.device = device
        self.features_fixed = False
        self.z_window_radius = 0.025
        self.inverse_v = None

    # --- N2: tet_bxfx4x3 from vertex positions (deftet.py:65-68)
    def gather_tet_pos(self, vertice_pos, tetrahedron_bxfx4):
        key = (tetrahedron_bxfx4.data_ptr(), tuple(tetrahedron_bxfx4.shape), tetrahedron_bxfx4._version, vertice_pos.shape[1])
        if getattr(self, "_topo_key", None) != key:
            self._topo = TetTopology(tetrahedron_bxfx4, vertice_pos.shape[1])
            self._topo_key = key
        return self._topo.gather(vertice_pos)

    # --- N1: GT occupancy of the tet centroids (deftet.py:33-49)
    def check_tet_inside_sdfs(self, tet_bxfx4x3, mesh_list):
        verts, faces = mesh_list[0], mesh_list[1]
        with torch.no_grad():
            center = torch.mean(tet_bxfx4x3, dim=2)                     # [B,T,3], same reduction as per shape
            same_mesh = all(f[0] is faces[0][0] for f in faces) and all(v.shape == verts[0].shape for v in verts)
            if same_mesh and len(verts) == tet_bxfx4x3.shape[0]:
                # one launch sequence for the whole batch (faces shared, per-shape vertices)
                v = torch.cat([x.reshape(1, -1, 3) for x in verts], dim=0)
                return hip_ops.check_sign(v, faces[0][0], center, check=False).unsqueeze(-1).float()
            # a different ground-truth mesh per shape (the reference loops over the batch): one ragged call
            return hip_ops.check_sign_ragged(list(verts), [f[0] for f in faces], center).unsqueeze(-1).float()

    # --- A7
    def get_boundary_index(self, tet_face_fx3, tet_idx_fx2, occ_bxn):
        return hip_ops.boundary_index(tet_face_fx3, tet_idx_fx2, occ_bxn, mode=1)

    def get_internal_index(self, tet_face_fx3, tet_idx_fx2, occ_bxn):
        return hip_ops.boundary_index(tet_face_fx3, tet_idx_fx2, occ_bxn, mode=2)

    # --- A1 / A1b
    def check_condition(self, tet_bxfx4x3, point_pos_bxpx3):
        return check_condition_f_base(tet_bxfx4x3, point_pos_bxpx3)

    def paste_occ(self, pred_tet_occ, condition):
        return paste_occ(pred_tet_occ, condition)

    # --- A11 (each call evaluates the fused kernel and returns its own component)
    def volume_variance(self, tet_bxfx4x3, base_area_mask=None, area_normalize=(20, 20), pow=2, center_occ=None):
        return hip_ops.tet_energies(tet_bxfx4x3, None, pow_v=pow, pow_e=self.pow)[:, 0]

    def amips_energy(self, tet_bxfx4x3, inverse_v, scale=20, center_occ=None, square=False):
        e = hip_ops.tet_energies(tet_bxfx4x3, inverse_v, pow_v=self.pow, pow_e=self.pow, scale=scale)[:, 1]
        if square:
            raise NotImplementedError("square=True is never used by the reference (deftet.py:286-287)")
        return e

    def edge_length(self, tet_bxfx4x3, pow=2):
        return hip_ops.tet_energies(tet_bxfx4x3, None, pow_v=self.pow, pow_e=pow, scale=20)[:, 2]

    def energies(self, tet_bxfx4x3, inverse_v):
        """(volume_variance, amips_energy, edge_length) from ONE fused evaluation — what
        forward_surface_align (:82-83,:105) needs per step."""
        out = hip_ops.tet_energies(tet_bxfx4x3, inverse_v, pow_v=self.pow, pow_e=self.pow, scale=20.0)
        return out[:, 0], out[:, 1], out[:, 2]

    # --- init-time helpers (pure torch, as in the reference)
    def my_inverse(self, T):
        det_m = (torch.abs(torch.det(T)) < 1e-10).float()                     # :215
        iden_m = torch.eye(T.shape[-1], dtype=torch.float, device=T.device).unsqueeze(0).expand(T.shape[0], -1, -1)
        tmp_m = T * (1 - det_m.unsqueeze(-1).unsqueeze(-1)) + iden_m * det_m.unsqueeze(-1).unsqueeze(-1)
        return torch.inverse(tmp_m), 1 - det_m

    def tet_inverse_v(self, init_tet_pos, init_tet_fx4, scale=20):
        vertice_pos = init_tet_pos.float()
        tet = vertice_pos[init_tet_fx4.long()]                                 # [T,4,3]
        A, B, C, D = (tet[:, i:i + 1, :] * scale for i in range(4))
        offset_vec = torch.cat([B - A, C - A, D - A], dim=1)                   # :316
        return self.my_inverse(offset_vec)[0]
