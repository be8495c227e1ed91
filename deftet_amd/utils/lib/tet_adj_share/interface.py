"""`Tet_adj_share().run(tet_list, n_point)` as utils/lib/tet_adj_share/interface.py:14-47 of the reference offers it:
int32 tets [T, 4] in, a list of four scipy COO matrices [T, T] out — entry (t, u) of matrix i is 1 when tet u shares
local face i of tet t.  The pairs come from `deftet_tet_adj_share_host` (GPU sort-based builder in libdeftet_hip.so,
row order of the reference's run.cpp) instead of utils/lib/tet_adj_share/run.so."""
import ctypes as c

import numpy as np
from scipy.sparse import coo_matrix

from deftet_amd.utils.lib import _host

_ENTRY = "deftet_tet_adj_share_host"


class Tet_adj_share:
    def __init__(self):
        self.run_native = _host.host_fn(_ENTRY, [_host.I32P, _host.I32P, _host.I32P, c.c_int, c.c_int])

    def run(self, tet_list, n_point):
        tets = _host.checked(tet_list, np.int32)
        n_tet = tets.shape[0]
        rows = _host.out_i32(8 * n_tet, 3)                  # (tet, neighbour, local face): at most two per face
        n_shared = _host.out_i32(1)                         # shared faces found; each contributes both directions
        _host.call(self.run_native, _ENTRY, _host.ptr(tets), _host.ptr(rows), _host.ptr(n_shared), int(n_point), n_tet)
        rows = rows[:2 * int(n_shared[0])]
        per_face = []
        for face in range(4):
            r = rows[rows[:, 2] == face]
            per_face.append(coo_matrix((np.ones(r.shape[0]), (r[:, 0], r[:, 1])), shape=(n_tet, n_tet)))
        return per_face
