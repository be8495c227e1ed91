"""ctypes binding of libdeftet_hip.so (the C ABI declared in include/deftet_hip.h).

The product path has NO CPU fallback: if the shared object is missing, or an op is handed
a non-GPU tensor, it raises.  Build the library with `python -m deftet_amd.build`
(or `__graft_entry__.build()`).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DEFTET_HIP_LIB") or os.path.join(HERE, "libdeftet_hip.so")   # env override: experiments only

_vp, _i, _sz, _ll, _f = C.c_void_p, C.c_int, C.c_size_t, C.c_longlong, C.c_float

# name -> (restype, argtypes); must list every symbol of include/deftet_hip.h
SIGNATURES = {
    "deftet_version": (_i, []),
    "deftet_bandwidth_probe": (_i, [_vp, _vp, _sz, _i, C.POINTER(_sz), _vp]),
    "deftet_last_error": (C.c_char_p, []),
    "deftet_device_count": (_i, []),
    "deftet_profile_select": (_i, [C.c_char_p]),
    "deftet_profile_read": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "deftet_point_in_tet_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "deftet_point_in_tet_hits_ints": (_sz, [_i, _i, _i]),
    "deftet_point_in_tet_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "deftet_point_in_tet_scan_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "deftet_tet_spatial_order_workspace_bytes": (_sz, [_i]),
    "deftet_tet_spatial_order_f32": (_i, [_vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "deftet_tet_order_coherence_workspace_bytes": (_sz, [_i]),
    "deftet_tet_order_coherence_f32": (_i, [_vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "deftet_point_in_tet_ex_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "deftet_point_in_tet_prepare_ex_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "deftet_point_in_tet_scan_ex_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "deftet_point_in_tet_prepare_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "deftet_point_in_tet_grid_dims": (_i, [_i, _i, C.POINTER(_i), C.POINTER(_i)]),
    "deftet_point_in_tet_resolve_algo": (_i, [_i, _i, _i]),
    "deftet_point_in_tet_read_stats": (_i, [_vp, _sz, _i, _i, _i, _i, _vp, _vp]),
    "deftet_point_in_tet_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "deftet_point_in_tet_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "deftet_paste_occ_fwd_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "deftet_paste_occ_bwd_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "deftet_check_sign_workspace_bytes": (_sz, [_i, _i, _i]),
    "deftet_check_sign_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "deftet_check_sign_ragged_workspace_bytes": (_sz, [_i, _ll, _i, _i]),
    "deftet_check_sign_ragged_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, _i, _i, _i, _vp, _sz, _vp]),
    "deftet_tet_edges_i64": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "deftet_subdivide_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "deftet_point_adj_table_i64": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "deftet_delete_tet_i64": (_i, [_vp, _vp, _f, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "deftet_tet_neighbour_weights_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "deftet_tet_gather_fwd_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "deftet_tet_vertex_csr_workspace_bytes": (_sz, [_i, _i, _i]),
    "deftet_tet_vertex_csr_i32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "deftet_tet_gather_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "deftet_put_host_ints": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "deftet_point_in_tet_bwd_to_vertices_workspace_bytes": (_sz, [_i, _i, _i]),
    "deftet_point_in_tet_bwd_to_vertices_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "deftet_rowdot_workspace_bytes": (_sz, [_i]),
    "deftet_rowdot_f32": (_i, [_vp, _vp, _vp, _i, _ll, _vp, _sz, _vp]),
    "deftet_rowdot2_f32": (_i, [_vp, _vp, _ll, _vp, _vp, _ll, _vp, _i, _vp, _sz, _vp]),
    "deftet_sqrt_rowsum_f32": (_i, [_vp, _f, _vp, _i, _ll, _vp, _sz, _vp]),
    "deftet_sqrt_rowsum_bwd_f32": (_i, [_vp, _f, _vp, _vp, _i, _ll, _vp]),
    "deftet_builder_workspace_bytes": (_sz, [_i, _i]),
    "deftet_tet_adj_share_i32": (_i, [_vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "deftet_tet_adj_share_host": (_i, [_vp, _vp, _vp, _i, _i]),
    "deftet_tet_face_adj_i32": (_i, [_vp, _vp, _ll, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "deftet_tet_face_adj_host": (_i, [_vp, _vp, _vp, _i, _i]),
    "deftet_tet_point_adj_i32": (_i, [_vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "deftet_tet_point_adj_host": (_i, [_vp, _vp, _vp, _i, _i]),
    "deftet_colaps_v_f32": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "deftet_colaps_v_host": (_i, [_vp, _vp, _vp, _vp, _i]),
    "deftet_tet_to_face_i32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "deftet_tet_neighbours_workspace_bytes": (_sz, [_i]),
    "deftet_tet_neighbours_i64": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "deftet_boundary_index_workspace_bytes": (_sz, [_i, _i]),
    "deftet_boundary_index_i64": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "deftet_tet_energies_workspace_bytes2": (_sz, [_i, _i]),
    "deftet_tet_energies_fwd_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _sz, _vp]),
    "deftet_tet_energies_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "deftet_face_edge_adj_workspace_bytes": (_sz, [_i]),
    "deftet_face_edge_adj_f32": (_i, [_vp, _vp, _i, _i, _vp, _sz, _vp]),
    "deftet_face_edge_adj_ragged_workspace_bytes": (_sz, [_i, _i]),
    "deftet_face_edge_adj_ragged_f32": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _sz, _vp]),
    "deftet_nn_index_ragged_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "deftet_normal_consistency_fwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "deftet_normal_consistency_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "deftet_tri_dist_workspace_bytes": (_sz, [_i, _i, _i]),
    "deftet_tri_dist_fwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "deftet_tri_dist_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "deftet_face_samples_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "deftet_chamfer_fwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "deftet_chamfer_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "deftet_tri_dist_fwd_order_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "deftet_tri_dist_bwd_order_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "deftet_nn_index_workspace_bytes": (_sz, [_i, _i, _i]),
    "deftet_nn_index_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "deftet_radix_sort_workspace_bytes": (_sz, [C.c_longlong, _i, _i]),
    "deftet_radix_sort": (_i, [_vp, _vp, _vp, _vp, C.c_longlong, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "deftet_scan_workspace_bytes": (_sz, [C.c_longlong, _i]),
    "deftet_scan": (_i, [_vp, _vp, C.c_longlong, _i, _i, _vp, _sz, _vp]),
    "deftet_sparse_render_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "deftet_sparse_render_fwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _sz, _vp]),
    "deftet_sparse_render_fwd_policy_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "deftet_sparse_render_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "deftet_sparse_render_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _sz, _vp]),
}

_lock = threading.Lock()
_lib = None


class DefTetHipError(RuntimeError):
    pass


def load():
    """Returns the loaded library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise DefTetHipError(
                    "libdeftet_hip.so is missing (%s). Build it with `python -m deftet_amd.build`; "
                    "there is no CPU fallback." % LIB_PATH)
            # torch bundles its own HIP runtime under the same SONAME (libamdhip64.so.7); it must
            # be the one already mapped when our library resolves its dependency, otherwise the
            # process ends up with two runtimes and ours sees no device.
            import torch  # noqa: F401
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                if os.environ.get("DEFTET_HIP_LIB") and not hasattr(lib, name):
                    continue                     # partial experiment builds only
                fn = getattr(lib, name)          # AttributeError = symbol not exported
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().deftet_last_error()
        raise DefTetHipError("%s failed (%d): %s" % (what, status, msg.decode() if msg else "?"))


# ------------------------------------------------------------------------------------
# torch plumbing: device pointers, current HIP stream, grow-only workspace per (device, stream)
# ------------------------------------------------------------------------------------
_ws = {}
_ws_lock = threading.Lock()


def require_gpu(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise DefTetHipError("deftet_amd operators need GPU tensors (got device %s); "
                                 "there is no CPU fallback" % t.device)


# Host-side cost of a call matters: a geometry step (bench.py --config 5) makes ~75 library calls between ~70 torch launches and
# was bound by the Python thread, not by the kernels (tools/probes/geometry_cpu_probe.py: 2.37 ms of host time in a 2.43 ms
# step; tools/probes/host_call_probe.py: 11-15 us per library call against 4 us per torch elementwise launch).  So: raw stream
# handles from torch._C (no Stream object), plain integers for pointers (ctypes converts them for c_void_p parameters), the
# device guard only when the current device is another one, the workspace looked up without the lock when it is large enough.
def ptr(t):
    return t.data_ptr() if t is not None else None


try:                                                    # (private, but stable since torch 1.x; the public route is ~6x slower)
    from torch._C import _cuda_getCurrentRawStream as _raw_stream, _cuda_getDevice as _cur_device
except Exception:                                       # pragma: no cover
    _raw_stream = _cur_device = None


def _dev_index(device):
    idx = device.index
    if idx is None:
        import torch
        idx = torch.cuda.current_device()
    return idx


def current_stream(device):
    """raw hipStream_t (int) of torch's current stream on `device`"""
    if _raw_stream is not None:
        return _raw_stream(_dev_index(device))
    import torch
    return torch.cuda.current_stream(device).cuda_stream


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(device):
    """`with on_device(dev):` == `with torch.cuda.device(dev):`, for free when dev is the current device already"""
    if _cur_device is not None and (device.index is None or device.index == _cur_device()):
        return _NO_GUARD
    import torch
    return torch.cuda.device(device)


def workspace(device, nbytes: int):
    """uint8 tensor of at least nbytes on `device`, cached per (device, stream)."""
    key = (_dev_index(device), current_stream(device))
    buf = _ws.get(key)                                  # (a dict read is atomic; the lock is for the grow path)
    if buf is not None and buf.numel() >= nbytes:
        return buf
    import torch
    with _ws_lock:
        buf = _ws.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=device)
            _ws[key] = buf
    return buf
