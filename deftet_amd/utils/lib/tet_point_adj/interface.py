"""Drop-in for /root/reference/utils/lib/tet_point_adj/interface.py:15-61:
`run(n_point, tet_list, normalize=False)` -> torch sparse [V,V] (values 1, or 1/deg(row)
when normalize); native call: deftet_tet_point_adj_host."""
import ctypes as c

import numpy as np
import torch

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
        idx = torch.from_numpy(edge[:n_edge[0], :].astype(np.int64))
        if normalize:
            # D^-1 A: every stored entry of row r weighs 1/deg(r) (reference interface.py:42-54 gets the same
            # numbers from two scipy products); entries are unique, so deg is a plain count
            deg = torch.bincount(idx[:, 0], minlength=int(n_point)).double()
            val = (1.0 / deg)[idx[:, 0]].float()
        else:
            val = torch.ones(idx.shape[0])
        return torch.sparse_coo_tensor(idx.t().contiguous(), val, torch.Size([int(n_point), int(n_point)]))
