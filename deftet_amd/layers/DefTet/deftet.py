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

    forward_surface_align / forward           (:51-184)   composed from the operators above plus the surface
                                                             terms in deftet_amd/surface_losses.py (A8/A9/A10); same
                                                             arguments, same tuple order as the reference
    laplacian_sparse                          (:340-343)  torch.sparse.mm (vendor SpMM)
"""
import collections
import threading
import torch
import torch.nn as nn

from deftet_amd import hip_ops, surface_losses
from deftet_amd.layers.DefTet.check_condition_tetrahedron_base.utils import check_condition_f_base, paste_occ, point_in_tet_occ_vertices

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

    _serials = __import__("itertools").count(1)

    def __init__(self, tet_idx, n_vertex):
        self.serial = next(TetTopology._serials)              # identifies the tet LIST: hip_ops.auto_tet_order keys its decision on it
        idx = tet_idx.long()
        # The reference hands every shape of a batch the SAME tet list (`tet_fx4.unsqueeze(0).expand(B, -1, -1)`,
        # train_multigpu.py:72-77): keep one copy then — the gather reads 8 MB of indices instead of 66 MB at res 70, B = 8,
        # the incidence CSR is built once instead of B times.  (One comparison + sync here, at build time only.)
        if idx.dim() == 3 and idx.shape[0] > 1 and bool((idx == idx[:1]).all()):
            idx = idx[:1]
        self.tet_idx = idx.contiguous()
        self.n_vertex = int(n_vertex)
        self.csr = hip_ops.tet_vertex_csr(self.tet_idx, self.n_vertex)

    def gather(self, vertice_pos):
        return _TetGather.apply(vertice_pos, self.tet_idx, self.csr)


# Cache of the incidence CSR, per device and index shape, at MODULE level: nn.DataParallel (train_multigpu.py:138)
# re-replicates the layer every step and discards whatever a replica stored on itself, and scatters a fresh copy of the
# index tensor.  Fast path: the very same tensor object, unmodified (a reference is kept, so its storage cannot be
# recycled for another tensor behind our back).  Otherwise the CONTENT is compared with the cached indices (one small
# kernel + sync; the CSR rebuild it avoids is 40x that) — never address or shape alone.
# nn.DataParallel runs the replicas' forward in threads of one process: the cache is shared by them, so every access is
# under a lock; least-recently-used entries go first (at most _TOPOLOGIES_MAX (device, shape, n_vertex) keys stay pinned in
# GPU memory; clear_topology_cache() drops them all).
_TOPOLOGIES = collections.OrderedDict()
_TOPOLOGIES_MAX = 16
_TOPOLOGIES_LOCK = threading.Lock()


def clear_topology_cache():
    with _TOPOLOGIES_LOCK:
        _TOPOLOGIES.clear()


def _topology_for(tetrahedron_bxfx4, n_vertex):
    key = (tetrahedron_bxfx4.device, tuple(tetrahedron_bxfx4.shape), int(n_vertex))
    with _TOPOLOGIES_LOCK:
        hit = _TOPOLOGIES.get(key)
    if hit is not None:
        topo, src, version = hit
        same = src is tetrahedron_bxfx4 and version == tetrahedron_bxfx4._version
        if not same:
            same = bool((tetrahedron_bxfx4 == topo.tet_idx).all())     # (broadcasts over the batch when one shared copy is kept)
        if same:
            with _TOPOLOGIES_LOCK:
                _TOPOLOGIES[key] = (topo, tetrahedron_bxfx4, tetrahedron_bxfx4._version)
                _TOPOLOGIES.move_to_end(key)
            return topo
    topo = TetTopology(tetrahedron_bxfx4, n_vertex)                   # (built outside the lock: a host sync and a few launches)
    with _TOPOLOGIES_LOCK:
        _TOPOLOGIES[key] = (topo, tetrahedron_bxfx4, tetrahedron_bxfx4._version)
        _TOPOLOGIES.move_to_end(key)
        while len(_TOPOLOGIES) > _TOPOLOGIES_MAX:
            _TOPOLOGIES.popitem(last=False)
    return topo


def save_tet_face(tri_fx3x3, f_name):
    """Triangle soup as an OBJ file: three `v` lines and one `f a c b` line per triangle, `%f` formatting
    (utils/mesh_utils.py:258-267)."""
    lines = []
    for i, t in enumerate(tri_fx3x3):
        for k in range(3):
            lines.append('v %f %f %f\n' % (t[k][0], t[k][1], t[k][2]))
        lines.append('f %d %d %d\n' % (i * 3 + 1, i * 3 + 3, i * 3 + 2))
    with open(f_name, 'w') as f:
        f.write(''.join(lines))


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
        return _topology_for(tetrahedron_bxfx4, vertice_pos.shape[1]).gather(vertice_pos)

    # --- N1: GT occupancy of the tet centroids (deftet.py:33-49)
    def check_tet_inside_sdfs(self, tet_bxfx4x3, mesh_list):
        verts, faces = mesh_list[0], mesh_list[1]
        with torch.no_grad():
            center = torch.mean(tet_bxfx4x3, dim=2)                     # [B,T,3], same reduction as per shape
            f0 = faces[0][0]                                            # f[0] makes a new view object each time: compare storage
            same_mesh = all(f[0].data_ptr() == f0.data_ptr() and f[0].shape == f0.shape and f[0].stride() == f0.stride()
                            for f in faces) and all(v.shape == verts[0].shape for v in verts)
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

    # --- A1 + A1b + paste_occ for a caller that holds the VERTICES (build-defined, like point_in_tet_occ): the gradient of the
    #     weights lands on vertice_pos without the dense per-tet gradient in between
    def occupancy_query(self, vertice_pos, tetrahedron_bxfx4, point_pos_bxpx3, pred_tet_occ, tet_bxfx4x3=None):
        """(condition [B,Q,1], weights [B,Q,4], occ [B,Q]) of the query points in the mesh (vertice_pos, tetrahedron_bxfx4);
        tet_bxfx4x3 = self.gather_tet_pos(vertice_pos, tetrahedron_bxfx4) when the caller has it already."""
        topo = _topology_for(tetrahedron_bxfx4, vertice_pos.shape[1])
        return point_in_tet_occ_vertices(vertice_pos, point_pos_bxpx3, pred_tet_occ, topo, tet_bxfx4x3)

    # --- A11 (each call evaluates the fused kernel and returns its own component)
    def volume_variance(self, tet_bxfx4x3, base_area_mask=None, area_normalize=(20, 20), pow=2, center_occ=None):
        return hip_ops.tet_energies(tet_bxfx4x3, None, pow_v=pow, pow_e=self.pow)[:, 0]

    def amips_energy(self, tet_bxfx4x3, inverse_v, scale=20, center_occ=None, square=False):
        if square:
            # never used by the reference's callers (deftet.py:286-287): the per-tet energies squared before the mean.  The
            # fused operator only hands out the means, so this option is the reference's own torch composition (on the GPU).
            n_batch = tet_bxfx4x3.shape[0]
            a, b, c, d = (tet_bxfx4x3[:, :, k, :].unsqueeze(2) * scale for k in range(4))
            offset_vec = torch.cat([b - a, c - a, d - a], dim=2)
            jac = torch.bmm(offset_vec.reshape(-1, 3, 3), inverse_v.unsqueeze(0).expand_as(offset_vec).reshape(-1, 3, 3))
            trace = (jac ** 2).sum(-1).sum(-1)
            det = (jac[:, 0, :] * torch.linalg.cross(jac[:, 1, :], jac[:, 2, :], dim=-1)).sum(-1)       # utils/matrix_utils.py:42-47
            energy = trace * torch.pow(det ** 2 + EPS, -1.0 / 3.0) * (det >= 0.0).float()
            return (energy.reshape(n_batch, -1) ** 2).mean(-1)
        return hip_ops.tet_energies(tet_bxfx4x3, inverse_v, pow_v=self.pow, pow_e=self.pow, scale=scale)[:, 1]

    def edge_length(self, tet_bxfx4x3, pow=2):
        return hip_ops.tet_energies(tet_bxfx4x3, None, pow_v=self.pow, pow_e=pow, scale=20)[:, 2]

    def energies(self, tet_bxfx4x3, inverse_v):
        """(volume_variance, amips_energy, edge_length) from ONE fused evaluation — what
        forward_surface_align (:82-83,:105) needs per step."""
        out = hip_ops.tet_energies(tet_bxfx4x3, inverse_v, pow_v=self.pow, pow_e=self.pow, scale=20.0)
        return out[:, 0], out[:, 1], out[:, 2]

    # --- the per-step composition (deftet.py:51-130): one fused launch sequence per operator for the whole
    #     batch; only the surface terms loop over shapes (a different predicted surface per shape)
    def forward_surface_align(self, vertice_pos, point_pos_bxpx3, tetrahedron_bxfx4=None, mesh_list=None,
                              gt_surface_points=None, tet_face_bxfx3=None, inference=False, pred_occ=None,
                              tet_face_tet_bx4fx2=None, save=False, save_name=None, inference_threshold=0.4, tet_bxfx4x3=None):
        # tet_bxfx4x3 (not in the reference's signature; its `forward` has the same optional argument, deftet.py:140): a
        # caller that has gathered the tet positions already — e.g. for the occupancy query of the same step — hands them
        # in instead of having them gathered (and their gradient scattered) a second time
        n_shape = vertice_pos.shape[0]
        if tet_bxfx4x3 is None:
            tet_bxfx4x3 = self.gather_tet_pos(vertice_pos, tetrahedron_bxfx4)
        center_occ = self.check_tet_inside_sdfs(tet_bxfx4x3, mesh_list)                       # [B,T,1], no grad
        face_fx3, face_tet_fx2 = tet_face_bxfx3[0], tet_face_tet_bx4fx2[0]
        boundary = self.get_boundary_index(face_fx3, face_tet_fx2, center_occ.squeeze(dim=-1))
        if save:
            # eval.py --save: the ground-truth-occupancy surface of the first five shapes as triangle-soup OBJ files
            # (deftet.py:72-80, utils/mesh_utils.py:258-267)
            for idx in range(min(len(boundary), 5)):
                tri = vertice_pos[idx][boundary[idx].long()]                               # [F,3,3]
                save_tet_face(tri.detach().cpu().numpy(), save_name + '_device_%d_%d.obj' % (torch.cuda.current_device(), idx))
        inv_v = self.inverse_v.to(tet_bxfx4x3.device)
        volume_variance, amips_energy, edge = self.energies(tet_bxfx4x3, inv_v)
        # The surface terms differ per shape (its own predicted boundary, a different face count).  The reference loops
        # over the shapes (deftet.py:89-103); here ONE ragged launch sequence covers the batch: the per-shape chains of
        # small launches would otherwise be bound by the host (8 shapes x ~150 framework + library calls ~ 25 ms).
        terms = surface_losses.surface_terms_batched(vertice_pos, boundary, gt_surface_points, per_face=20, stacked=True,
                                                     uv=getattr(self, "sample_uv", None))             # [3,B] (sample_uv: tests only)
        sum_chamfer, sum_analytic, sum_normal = terms.mean(1, keepdim=True)                          # three [1] tensors (:104-110)
        center_occ = center_occ.squeeze(-1)
        if inference:
            assert point_pos_bxpx3 is not None, 'point_pos_bxpx3 not given'
            # (the traversal order of the query is decided per TOPOLOGY when the index list is at hand; positions only otherwise)
            topo = _topology_for(tetrahedron_bxfx4, vertice_pos.shape[1]) if tetrahedron_bxfx4 is not None else None
            condition = check_condition_f_base(tet_bxfx4x3, point_pos_bxpx3, topo)
            pred_surface_face = self.get_boundary_index(face_fx3, face_tet_fx2, (pred_occ > inference_threshold).float())
            return (amips_energy, edge, volume_variance, sum_analytic, sum_normal, center_occ, condition, boundary,
                    pred_surface_face, sum_chamfer)
        return (amips_energy, edge, volume_variance, sum_analytic, sum_normal, center_occ, boundary, sum_chamfer,
                torch.zeros_like(sum_normal))

    def forward(self, v_pos_bxnx3=None, tet_bxfx4=None, boundary_bxfx3=None, gt_surface_point=None, inverse_offset=None,
                tet_bxfx4x3=None, calculate_amips_volume=True):
        """(chamfer, analytic, normal[, volume variance, amips, tet positions]) of one predicted surface
        (deftet.py:138-184); all ones when the surface is empty (:159-163)."""
        extra = ()
        if calculate_amips_volume:
            tet_pos = tet_bxfx4x3 if tet_bxfx4x3 is not None else self.gather_tet_pos(v_pos_bxnx3, tet_bxfx4)
            if inverse_offset is not None:
                vv, am, _ = self.energies(tet_pos, inverse_offset.to(tet_pos.device))
            else:
                vv = self.volume_variance(tet_pos, pow=self.pow)
                am = torch.zeros_like(vv)
            extra = (vv, am, tet_pos)
        if boundary_bxfx3.shape[1] == 0:
            one = torch.ones(1, device=boundary_bxfx3.device)
            return (one, one, one) + extra
        return surface_losses.surface_terms(v_pos_bxnx3, boundary_bxfx3, gt_surface_point, per_face=20) + extra

    def laplacian_sparse(self, offset, adj):
        """sum over vertices and coordinates of (mean of the neighbours' offsets - own offset)^2, per shape;
        adj = row-normalised vertex adjacency [V,V] (c_tet_to_adj_sparse(normalize=True))."""
        n_shape, n_vertex, width = offset.shape
        flat = offset.permute(1, 0, 2).reshape(n_vertex, n_shape * width)
        nei = torch.sparse.mm(adj, flat).reshape(n_vertex, n_shape, width).permute(1, 0, 2)
        return ((nei - offset) ** 2).sum(dim=(1, 2))

    # --- init-time helpers (pure torch, as in the reference)
    def my_inverse(self, T):
        det_m = (torch.abs(torch.det(T)) < 1e-10).float()                     # :215
        iden_m = torch.eye(T.shape[-1], dtype=torch.float, device=T.device).unsqueeze(0).expand(T.shape[0], -1, -1)
        tmp_m = T * (1 - det_m.unsqueeze(-1).unsqueeze(-1)) + iden_m * det_m.unsqueeze(-1).unsqueeze(-1)
        return torch.inverse(tmp_m).contiguous(), 1 - det_m              # (row-major: the energies would copy it per call)

    def tet_inverse_v(self, init_tet_pos, init_tet_fx4, scale=20):
        vertice_pos = init_tet_pos.float()
        tet = vertice_pos[init_tet_fx4.long()]                                 # [T,4,3]
        A, B, C, D = (tet[:, i:i + 1, :] * scale for i in range(4))
        offset_vec = torch.cat([B - A, C - A, D - A], dim=1)                   # :316
        return self.my_inverse(offset_vec)[0]
