"""Front-ends with the names of /root/reference/utils/tet_utils.py for the builders that
run in libdeftet_hip.so: c_tet_to_adj_sparse (:94-95), c_tet_to_face_adj_sparse (:203-205),
c_tet_adj_share (:371-375) and the live face-table builder tet_to_face (:208-256), plus
the render-side tet_to_face_idx (diff_render/diftet_6_subdiv/3_model/prepare_for_wz.py:49-104).

Unlike the reference module, importing this one does not dlopen anything from os.getcwd()
(utils/tet_utils.py:20-22): the interface objects are created on first use.
"""
import numpy as np
import torch

from deftet_amd import hip_ops

_objs = {}


def _obj(name):
    if name not in _objs:
        if name == "point":
            from deftet_amd.utils.lib.tet_point_adj.interface import Tet_point_adj as C
        elif name == "face":
            from deftet_amd.utils.lib.tet_face_adj.interface import Tet_face_adj as C
        else:
            from deftet_amd.utils.lib.tet_adj_share.interface import Tet_adj_share as C
        _objs[name] = C()
    return _objs[name]


def convert_torch_sparse(adj):
    """utils/matrix_utils.py:14-20"""
    idx = np.stack([adj.row, adj.col], axis=0)
    return torch.sparse_coo_tensor(torch.from_numpy(idx).long(), torch.from_numpy(adj.data).float(), adj.shape)


def c_tet_to_adj_sparse(points, tet_list, normalize=True):
    return _obj("point").run(points.shape[0], np.asarray(tet_list).astype(np.int32), normalize)


def c_tet_to_face_adj_sparse(points, tet_list):
    return _obj("face").run(points.shape[0], np.asarray(tet_list).astype(np.int32))


def c_tet_adj_share(tet_list, n_point, torch_t=True):
    adj_list = _obj("share").run(np.asarray(tet_list).astype(np.int32), n_point)
    if torch_t:
        adj_list = [convert_torch_sparse(adj) for adj in adj_list]
    return adj_list


def tet_to_face(n_point, tet_list, device="cuda"):
    """Same return tuple as utils/tet_utils.py:208-256: (tet_face_fx3, tet_face_tetidx_fx2,
    tet_face_tetfaceidx_fx2, tet_boundary_face) as numpy int64 arrays, first-seen order."""
    f3, t2, tf2, b3, n_multi = hip_ops.tet_to_face(np.asarray(tet_list), n_point, device, with_boundary=False)
    print('Cnt neighbor tet: ', [int(b3.shape[0]), int(f3.shape[0]), int(n_multi)])      # tet_utils.py:255
    return f3.cpu().numpy(), t2.cpu().numpy(), tf2.cpu().numpy(), b3.cpu().numpy()


def tet_to_face_idx(n_point, tet_list, with_boundary=False, device="cuda"):
    """prepare_for_wz.py:49-104: boundary faces inline with partner -1 when with_boundary."""
    f3, t2, tf2, b3, n_multi = hip_ops.tet_to_face(np.asarray(tet_list), n_point, device, with_boundary=with_boundary)
    return f3.cpu().numpy(), t2.cpu().numpy(), tf2.cpu().numpy()


def tet_to_face_withtet(points, tet_list, device="cuda"):
    """utils/tet_utils.py:259-300: int64 [4T,2], row 4t+i = the owners of local face i's key in insertion
    order, a lone owner padded with 0."""
    return hip_ops.tet_neighbours(np.asarray(tet_list), points.shape[0], device, want_face_owners=True)[1].cpu().numpy()


def tet_neighbour_table(tet_list_tx4, n_point, device="cuda"):
    """The T x 4 `tet_neighbour_idx` that diff_render/diftet_6_subdiv/3_model/utils_tetsv.py:16-75 returns next to
    the four adjacency matrices (consumed at 3_model/deftet.py:152,327)."""
    return hip_ops.tet_neighbours(np.asarray(tet_list_tx4), n_point, device).cpu().numpy()
