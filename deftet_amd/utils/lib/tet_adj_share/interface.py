"""Drop-in for /root/reference/utils/lib/tet_adj_share/interface.py:14-47: same class name,
`run(tet_list, n_point)` signature, buffer sizing and return value (four scipy COO T x T
matrices, one per local face id); the native call goes to libdeftet_hip.so's
deftet_tet_adj_share_host (GPU sort-based builder) instead of utils/lib/tet_adj_share/run.so."""
import ctypes as c

import numpy as np
from scipy.sparse import coo_matrix

from deftet_amd.utils.lib import _host


class Tet_adj_share:
    def __init__(self):
        self.run_native = _host.host_fn("deftet_tet_adj_share_host", [_host.I32P, _host.I32P, _host.I32P, c.c_int, c.c_int])

    def run(self, tet_list, n_point):
        assert tet_list.dtype == np.int32
        tet_list = np.ascontiguousarray(tet_list)
        tet_list_p = tet_list.ctypes.data_as(_host.I32P)
        n_face = tet_list.shape[0] * 4
        index_list = np.zeros((n_face * 2, 3), dtype=np.int32)
        n_face_edge = np.zeros(1, dtype=np.int32)
        _host.call(self.run_native, "deftet_tet_adj_share_host", tet_list_p, index_list.ctypes.data_as(_host.I32P),
                   n_face_edge.ctypes.data_as(_host.I32P), int(n_point), tet_list.shape[0])
        n_tet = tet_list.shape[0]
        index_list = index_list[:n_face_edge[0] * 2]
        index_value = np.ones(index_list.shape[0])
        adj_list = []
        for i in range(4):
            sel = index_list[:, 2] == i
            adj_list.append(coo_matrix((index_value[sel], (index_list[:, 0][sel], index_list[:, 1][sel])),
                                       shape=(n_tet, n_tet)))
        return adj_list
