"""Thin torch-tensor front-ends over the C ABI (one function per entry point).

Everything here is plumbing: argument checks, output allocation, workspace, stream.  The
reference-shaped operator surface (torch.autograd.Function / nn.Module with the
reference's names and signatures) lives in deftet_amd/layers/ and deftet_amd/utils/.
"""
from __future__ import annotations

import torch

from . import _lib

PIT_AUTO, PIT_BRUTE = 0, 1


def _f32c(t):
    return t.contiguous().float()


# --------------------------------------------------------------------------------- A1
def point_in_tet(tet_bxtx4x3, pts_bxqx3, want_bary=False, algo=PIT_AUTO):
    """cond f32 [B,Q,1] (lowest containing tet index or -1) and optionally the
    barycentric weights f32 [B,Q,4] of the hit tet."""
    _lib.require_gpu(tet_bxtx4x3, pts_bxqx3)
    lib = _lib.load()
    tet, pts = _f32c(tet_bxtx4x3), _f32c(pts_bxqx3)
    if tet.dim() != 4 or tet.shape[2:] != (4, 3):
        raise RuntimeError("tet_bxfx4x3 must be [B,T,4,3], got %s" % (tuple(tet.shape),))
    if pts.dim() != 3 or pts.shape[2] != 3 or pts.shape[0] != tet.shape[0]:
        raise RuntimeError("point_pos_bxnx3 must be [B,Q,3] with the same B, got %s" % (tuple(pts.shape),))
    B, T, Q = tet.shape[0], tet.shape[1], pts.shape[1]
    dev = pts.device
    cond = torch.empty(B, Q, 1, device=dev, dtype=torch.float32)
    bary = torch.empty(B, Q, 4, device=dev, dtype=torch.float32) if want_bary else None
    with torch.cuda.device(dev):
        nbytes = lib.deftet_point_in_tet_workspace_bytes(B, T, Q, algo)
        ws = _lib.workspace(dev, nbytes)
        _lib.check(lib.deftet_point_in_tet_f32(_lib.ptr(tet), _lib.ptr(pts), _lib.ptr(cond), _lib.ptr(bary), B, T, Q,
                                               algo, _lib.ptr(ws), ws.numel(), _lib.current_stream(dev)),
                   "deftet_point_in_tet_f32")
    return (cond, bary) if want_bary else cond


def point_in_tet_bwd(tet_bxtx4x3, pts_bxqx3, cond, grad_w, want_grad_pts=False):
    _lib.require_gpu(tet_bxtx4x3, pts_bxqx3, cond, grad_w)
    lib = _lib.load()
    tet, pts, cond, gw = _f32c(tet_bxtx4x3), _f32c(pts_bxqx3), _f32c(cond), _f32c(grad_w)
    B, T, Q = tet.shape[0], tet.shape[1], pts.shape[1]
    dev = pts.device
    grad_tet = torch.empty_like(tet)
    grad_pts = torch.empty_like(pts) if want_grad_pts else None
    with torch.cuda.device(dev):
        ws = _lib.workspace(dev, lib.deftet_point_in_tet_bwd_workspace_bytes(B, T, Q))
        _lib.check(lib.deftet_point_in_tet_bwd_f32(_lib.ptr(tet), _lib.ptr(pts), _lib.ptr(cond), _lib.ptr(gw),
                                                   _lib.ptr(grad_tet), _lib.ptr(grad_pts), B, T, Q, 0,
                                                   _lib.ptr(ws), ws.numel(), _lib.current_stream(dev)),
                   "deftet_point_in_tet_bwd_f32")
    return grad_tet, grad_pts


def paste_occ_fwd(pred_bxt, cond_bxqx1, clamp_inplace=True):
    _lib.require_gpu(pred_bxt, cond_bxqx1)
    lib = _lib.load()
    pred = _f32c(pred_bxt)
    if not (cond_bxqx1.is_contiguous() and cond_bxqx1.dtype == torch.float32):
        raise RuntimeError("condition must be a contiguous float32 tensor (it is clamped in place)")
    B, T = pred.shape
    Q = cond_bxqx1.shape[1]
    out = torch.empty(B, Q, device=pred.device, dtype=torch.float32)
    with torch.cuda.device(pred.device):
        _lib.check(lib.deftet_paste_occ_fwd_f32(_lib.ptr(pred), _lib.ptr(cond_bxqx1), _lib.ptr(out), B, T, Q,
                                                int(clamp_inplace), _lib.current_stream(pred.device)),
                   "deftet_paste_occ_fwd_f32")
    return out


def paste_occ_bwd(cond_bxqx1, grad_out_bxq, n_tet):
    _lib.require_gpu(cond_bxqx1, grad_out_bxq)
    lib = _lib.load()
    cond, go = _f32c(cond_bxqx1), _f32c(grad_out_bxq)
    B, Q = go.shape
    gp = torch.empty(B, n_tet, device=go.device, dtype=torch.float32)
    with torch.cuda.device(go.device):
        _lib.check(lib.deftet_paste_occ_bwd_f32(_lib.ptr(cond), _lib.ptr(go), _lib.ptr(gp), B, n_tet, Q, 1,
                                                _lib.current_stream(go.device)), "deftet_paste_occ_bwd_f32")
    return gp
