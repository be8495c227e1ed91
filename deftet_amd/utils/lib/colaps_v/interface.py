"""Drop-in for /root/reference/utils/lib/colaps_v/interface.py:15-35 (the reference names this
class Tet_point_adj as well — kept, with a clearer alias): `run(point_nx3)` ->
(map_array int32 [N], inverse_idx int32 [k]); native call: deftet_colaps_v_host."""
import ctypes as c

import numpy as np

from deftet_amd.utils.lib import _host


class Tet_point_adj:
    def __init__(self):
        self.run_native = _host.host_fn("deftet_colaps_v_host", [_host.F32P, _host.I32P, _host.I32P, _host.I32P, c.c_int])

    def run(self, point_nx3):
        assert point_nx3.dtype == np.float32
        point = np.ascontiguousarray(point_nx3)
        n_point = point.shape[0]
        map_array = np.zeros(n_point, dtype=np.int32)
        inverse_idx = np.zeros(n_point, dtype=np.int32)
        n_colaps_v = np.zeros(1, dtype=np.int32)
        _host.call(self.run_native, "deftet_colaps_v_host", point.ctypes.data_as(_host.F32P),
                   map_array.ctypes.data_as(_host.I32P), inverse_idx.ctypes.data_as(_host.I32P),
                   n_colaps_v.ctypes.data_as(_host.I32P), n_point)
        return map_array, inverse_idx[:n_colaps_v[0]]


Colaps_v = Tet_point_adj
