"""Drop-in for /root/reference/utils/lib/tet_face_adj/interface.py:15-37: `run(n_point,
tet_list)` -> scipy CSR (4T x 4T) of ones; native call: deftet_tet_face_adj_host (same
32-bit edge-key behaviour as run.cpp:39)."""
import ctypes as c

import numpy as np
from scipy.sparse import coo_matrix

from deftet_amd.utils.lib import _host


class Tet_face_adj:
    def __init__(self):
        self.run_native = _host.host_fn("deftet_tet_face_adj_host", [_host.I32P, _host.I32P, _host.I32P, c.c_int, c.c_int])

    def run(self, n_point, tet_list):
        assert tet_list.dtype == np.int32
        tet_list = np.ascontiguousarray(tet_list)
        n_face = tet_list.shape[0] * 4
        face_edge = np.zeros((n_face * 50, 2), dtype=np.int32)
        n_face_edge = np.zeros(1, dtype=np.int32)
        _host.call(self.run_native, "deftet_tet_face_adj_host", tet_list.ctypes.data_as(_host.I32P),
                   face_edge.ctypes.data_as(_host.I32P), n_face_edge.ctypes.data_as(_host.I32P), int(n_point),
                   tet_list.shape[0])
        n = int(n_face_edge[0])
        v = np.ones(n)
        return coo_matrix((v, (face_edge[:n, 0], face_edge[:n, 1])), shape=(n_face, n_face)).tocsr()
