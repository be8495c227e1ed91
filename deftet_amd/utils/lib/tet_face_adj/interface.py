"""`Tet_face_adj().run(n_point, tet_list)` as utils/lib/tet_face_adj/interface.py:15-37 of the reference: int32 tets
[T, 4] in, scipy CSR [4T, 4T] of ones out (faces of the tet soup that share an edge).  Native entry:
`deftet_tet_face_adj_host` (the 32-bit edge-key behaviour of run.cpp:39 included)."""
import ctypes as c

import numpy as np
from scipy.sparse import coo_matrix

from deftet_amd.utils.lib import _host

_ENTRY = "deftet_tet_face_adj_host"
_PAIRS_PER_FACE = 50                                        # the reference's output capacity per face


class Tet_face_adj:
    def __init__(self):
        self.run_native = _host.host_fn(_ENTRY, [_host.I32P, _host.I32P, _host.I32P, c.c_int, c.c_int])

    def run(self, n_point, tet_list):
        tets = _host.checked(tet_list, np.int32)
        n_face = 4 * tets.shape[0]
        pairs, n_pairs = _host.out_i32(n_face * _PAIRS_PER_FACE, 2), _host.out_i32(1)
        _host.call(self.run_native, _ENTRY, _host.ptr(tets), _host.ptr(pairs), _host.ptr(n_pairs), int(n_point), tets.shape[0])
        pairs = pairs[:int(n_pairs[0])]
        return coo_matrix((np.ones(pairs.shape[0]), (pairs[:, 0], pairs[:, 1])), shape=(n_face, n_face)).tocsr()
