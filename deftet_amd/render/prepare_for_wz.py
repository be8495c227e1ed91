"""HIP-backed twins of the geometry rebuilds in
/root/reference/diff_render/diftet_6_subdiv/3_model/prepare_for_wz.py (same names, argument order
and return values).  The reference works on numpy arrays on the host; these accept numpy arrays
(returned as numpy, like the reference) or GPU torch tensors (returned as tensors, no copies).

    generate_edge            (:184-203)   generate_tet_edge_idx  (:223-236)
    generate_edge_points     (:238-252)   generate_subdivision   (:255-301)
    generate_point_adj_idx   (:134-146)   delete_tet             (:171-180)
    tetweights2tetneighbourweights (3_model/deftet.py:316-331)
"""
import numpy as np
import torch

from deftet_amd import hip_ops


def _dev():
    if not torch.cuda.is_available():
        raise hip_ops._lib.DefTetHipError("deftet_amd operators need a GPU; there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _in(x, dtype=None):
    """-> (GPU tensor, was_numpy)"""
    if isinstance(x, torch.Tensor):
        return (x if dtype is None else x.to(dtype)), False
    t = torch.from_numpy(np.ascontiguousarray(x))
    return (t if dtype is None else t.to(dtype)).to(_dev()), True


def _out(t, as_numpy):
    return t.cpu().numpy() if as_numpy else t


def generate_edge(tet_list_tx4):
    tet, npy = _in(tet_list_tx4, torch.int64)
    n_point = int(tet.max().item()) + 1 if tet.numel() else 0
    edges, _ = hip_ops.tet_edges(tet, n_point)
    return _out(edges, npy)


def generate_tet_edge_idx(tet_list_tx4, edges_all_ex2):
    """edges_all_ex2 must be generate_edge(tet_list_tx4) (it is in the reference); it is recomputed."""
    tet, npy = _in(tet_list_tx4, torch.int64)
    n_point = int(tet.max().item()) + 1 if tet.numel() else 0
    _, tet_edge = hip_ops.tet_edges(tet, n_point)
    return _out(tet_edge, npy)


def generate_edge_points(tet_points_px3, tet_feat_pxk, edges_all_ex2):
    pts, npy = _in(tet_points_px3, torch.float32)
    feat, _ = _in(tet_feat_pxk, torch.float32)
    edges, _ = _in(edges_all_ex2, torch.int64)
    mid_p = (pts[edges[:, 0]] + pts[edges[:, 1]]) / 2
    mid_f = (feat[edges[:, 0]] + feat[edges[:, 1]]) / 2
    return _out(mid_p, npy), _out(mid_f, npy)


def generate_subdivision(tet_list_tx4, tet_points_px3, tet_feat_pxk, tet_list_subdiv_sig=None):
    tet, npy = _in(tet_list_tx4, torch.int64)
    pts, _ = _in(tet_points_px3, torch.float32)
    feat, _ = _in(tet_feat_pxk, torch.float32)
    sig = None if tet_list_subdiv_sig is None else _in(tet_list_subdiv_sig, torch.bool)[0]
    pn, fn, tn = hip_ops.subdivide(tet, pts, feat, sig)
    return _out(pn, npy), _out(fn, npy), _out(tn, npy)


def generate_point_adj_idx(n_point, tet_list):
    tet, npy = _in(tet_list, torch.int64)
    table, adjsum = hip_ops.point_adj_idx(n_point, tet)
    return _out(table, npy), _out(adjsum, npy)


def delete_tet(tet_list_tx4, tet_weights_tx4, thres=0.01):
    tet, npy = _in(tet_list_tx4, torch.int64)
    w, _ = _in(tet_weights_tx4, torch.float32)
    return _out(hip_ops.delete_tet(tet, w, thres), npy)


def tetweights2tetneighbourweights(tet_weights_tx4, tet_neighbour_idx, neilevel=1):
    """The reference method reads self.tet_neighbour_idx; here it is the second argument."""
    w, npy = _in(tet_weights_tx4, torch.float32)
    nei, _ = _in(tet_neighbour_idx, torch.int64)
    return _out(hip_ops.tet_neighbour_weights(w, nei.reshape(-1, 4), neilevel), npy)
