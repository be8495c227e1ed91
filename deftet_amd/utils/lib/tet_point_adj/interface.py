"""Drop-in for /root/reference/utils/lib/tet_point_adj/interface.py:15-61:
`run(n_point, tet_list, normalize=False)` -> torch sparse [V,V] (values 1, or 1/deg(row)
when normalize); native call: deftet_tet_point_adj_host."""
import ctypes as c

import numpy as np
import torch
from scipy.sparse import coo_matrix

from deftet_amd.utils.lib import _host


class Tet_point_adj:
    def __init__(self):
        self.run_native = _host.host_fn("deftet_tet_point_adj_host", [_host.I32P, _host.I32P, _host.I32P, c.c_int, c.c_int])

    def run(self, n_point, tet_list, normalize=False):
        assert tet_list.dtype == np.int32
        tet_list = np.ascontiguousarray(tet_list)
        edge = np.zeros((tet_list.shape[0] * 12, 2), dtype=np.int32)
        n_edge = np.zeros(1, dtype=np.int32)
        _host.call(self.run_native, "deftet_tet_point_adj_host", tet_list.ctypes.data_as(_host.I32P),
                   edge.ctypes.data_as(_host.I32P), n_edge.ctypes.data_as(_host.I32P), int(n_point), tet_list.shape[0])
        idx = edge[:n_edge[0], :]
        v = np.ones(idx.shape[0])
        if normalize:                                              # interface.py:42-54
            adj_m = coo_matrix((v, (idx[:, 0], idx[:, 1])), shape=(n_point, n_point))
            sum_adj = 1.0 / adj_m.sum(axis=-1)
            n_point = sum_adj.shape[0]
            new_idx = list(range(n_point))
            sum_m = coo_matrix((np.asarray(sum_adj).reshape(-1), (new_idx, new_idx)), shape=(n_point, n_point))
            adj = sum_m.dot(adj_m)
            idx = np.asarray(adj.nonzero())
            return torch.sparse_coo_tensor(torch.from_numpy(idx).long(), torch.from_numpy(adj.data).float(),
                                           torch.Size([n_point, n_point]))
        idx = torch.from_numpy(idx.astype(np.int64))
        return torch.sparse_coo_tensor(idx.transpose(0, 1), torch.ones(idx.shape[0]), torch.Size([n_point, n_point]))
