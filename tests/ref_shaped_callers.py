"""Callers written the way the reference's own modules reach the hot path — same import statements,
same call shapes — used to prove that `deftet_amd.overlay.install()` is a drop-in:

    layers/DefTet/deftet.py:12-16      from utils import ...; import kaolin as kal;
                                       from layers.DefTet.check_condition_tetrahedron_base.utils import check_condition_f_base
    utils/mesh_utils.py:11-13          from layers.DefTet.tet_face_adj_m_idx.utils import tet_face_adj_m_f_idx
                                       from layers.nearest_neighbor import NearestNeighbor
                                       from layers.DefTet.tet_analytic_distance_batch.utils import tet_analytic_distance_f_batch
    utils/tet_utils.py:15-22           from utils.lib.tet_point_adj.interface import Tet_point_adj  (etc.)
    5_rendereq/deftetrneder.py:24      import kaolin as kal  ->  kal.render.mesh.deftet_sparse_render(...)

Import this module only AFTER the overlay is installed."""


def import_like_reference():
    import kaolin as kal
    from layers.DefTet.check_condition_tetrahedron_base.utils import check_condition_f_base
    from layers.DefTet.tet_face_adj_m_idx.utils import tet_face_adj_m_f_idx
    from layers.DefTet.tet_analytic_distance_batch.utils import tet_analytic_distance_f_batch
    from layers.nearest_neighbor import NearestNeighbor
    from utils.lib.tet_point_adj.interface import Tet_point_adj
    from utils.lib.tet_face_adj.interface import Tet_face_adj
    from utils.lib.tet_adj_share.interface import Tet_adj_share
    return dict(kal=kal, check_condition_f_base=check_condition_f_base, tet_face_adj_m_f_idx=tet_face_adj_m_f_idx,
                tet_analytic_distance_f_batch=tet_analytic_distance_f_batch, NearestNeighbor=NearestNeighbor,
                Tet_point_adj=Tet_point_adj, Tet_face_adj=Tet_face_adj, Tet_adj_share=Tet_adj_share)


def occupancy_of_centroids(tet_bxfx4x3, verts_list, faces_list):
    """the loop of DefTet.check_tet_inside_sdfs (deftet.py:33-49) against `kal.ops.mesh.check_sign`"""
    import torch
    import kaolin as kal
    out = []
    for v, f, tet in zip(verts_list, faces_list, tet_bxfx4x3):
        centre = tet.mean(dim=1)
        out.append(kal.ops.mesh.check_sign(v, f[0], centre.unsqueeze(0), hash_resolution=512).unsqueeze(-1))
    return torch.cat(out, dim=0).float()


def query_and_paste(tet_bxfx4x3, points_bxpx3, pred_tet_occ):
    """eval-time path (deftet.py:112 + train_multigpu.py:383): condition, then the paste_occ gather"""
    import torch
    from layers.DefTet.check_condition_tetrahedron_base.utils import check_condition_f_base
    condition = check_condition_f_base(tet_bxfx4x3, points_bxpx3)
    c = condition.clone()
    c[c < 0] = 0
    return condition, torch.gather(input=pred_tet_occ, index=c.long().squeeze(-1), dim=1)


def vertex_adjacency(n_point, tets_int32, normalize=True):
    """train_multigpu.py:72 -> utils/tet_utils.py:94-95 -> Tet_point_adj().run"""
    from utils.lib.tet_point_adj.interface import Tet_point_adj
    return Tet_point_adj().run(n_point, tets_int32, normalize=normalize)
