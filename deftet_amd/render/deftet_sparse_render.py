"""Differentiable tet-face rasterizer with the call signature of
`kaolin.render.mesh.deftet_sparse_render` as the reference uses it
(/root/reference/diff_render/diftet_6_subdiv/5_rendereq/deftetrneder.py:97-100).  The host code
around the call (`peel2mask`, `vertex2face`, `perspective`) stays the reference's own under the
INTEGRATION.md overlay; `deftet_amd/render/compositing.py` holds this repository's compositing step.

PARITY UNPINNED for the rasterizer itself: Kaolin is not part of the reference tree and the
reference does not pin a version (README.md:30); see deftet_amd/csrc/raster.hip for the exact
arithmetic this build implements.
"""
import torch
from torch.autograd import Function

from deftet_amd import _lib


NEAREST, FIRST = 0, 1     # saturation policy (include/deftet_hip.h: DEFTET_RASTER_NEAREST / DEFTET_RASTER_FIRST)


def _f32c(t):
    return t.contiguous().float()


class _SparseRender(Function):
    @staticmethod
    def forward(ctx, pixel_coords, render_ranges, face_vertices_z, face_vertices_image, face_features, knum, eps, policy):
        _lib.require_gpu(pixel_coords, render_ranges, face_vertices_z, face_vertices_image, face_features)
        lib = _lib.load()
        pix, rng = _f32c(pixel_coords), _f32c(render_ranges)
        fz, fxy, ff = _f32c(face_vertices_z), _f32c(face_vertices_image), _f32c(face_features)
        B, P = pix.shape[0], pix.shape[1]
        F, D = fz.shape[1], ff.shape[3]
        if fxy.shape != (B, F, 3, 2) or ff.shape[:3] != (B, F, 3) or rng.shape != (B, P, 2):
            raise RuntimeError("deftet_sparse_render: inconsistent shapes")
        dev = pix.device
        feat = torch.empty(B, P, knum, D, device=dev, dtype=torch.float32)
        face = torch.empty(B, P, knum, device=dev, dtype=torch.int64)
        with _lib.on_device(dev):
            ws = _lib.workspace(dev, lib.deftet_sparse_render_workspace_bytes(B, P, F, knum))
            _lib.check(lib.deftet_sparse_render_fwd_policy_f32(_lib.ptr(pix), _lib.ptr(rng), _lib.ptr(fz), _lib.ptr(fxy), _lib.ptr(ff),
                                                               _lib.ptr(feat), _lib.ptr(face), _lib.ptr(None), B, P, F, D, knum, eps,
                                                               int(policy), _lib.ptr(ws), ws.numel(), _lib.current_stream(dev)),
                       "deftet_sparse_render_fwd_policy_f32")
        ctx.save_for_backward(pix, fxy, ff, face)
        ctx.eps = eps
        ctx.mark_non_differentiable(face)
        # (no zero tensors for outputs nobody differentiated: autograd would otherwise fill an int64 [B,P,knum] "gradient" for `face`
        # on every backward — 134 MB, 23 us at BASELINE configs[4])
        ctx.set_materialize_grads(False)
        return feat, face

    @staticmethod
    def backward(ctx, grad_feat, _grad_face):
        pix, fxy, ff, face = ctx.saved_tensors
        if grad_feat is None:                                  # nothing flowed into the features: no gradient for anybody
            return None, None, None, None, None, None, None, None
        lib = _lib.load()
        B, P, knum = face.shape
        F, D = fxy.shape[1], ff.shape[3]
        g = _f32c(grad_feat)
        gxy = torch.empty_like(fxy)
        gff = torch.empty_like(ff)
        dev = pix.device
        with _lib.on_device(dev):
            ws = _lib.workspace(dev, lib.deftet_sparse_render_bwd_workspace_bytes(B, P, F, knum))
            _lib.check(lib.deftet_sparse_render_bwd_f32(_lib.ptr(pix), _lib.ptr(fxy), _lib.ptr(ff), _lib.ptr(face), _lib.ptr(None),
                                                        _lib.ptr(g), _lib.ptr(gxy), _lib.ptr(gff), B, P, F, D, knum, ctx.eps,
                                                        _lib.ptr(ws), ws.numel(), _lib.current_stream(dev)),
                       "deftet_sparse_render_bwd_f32")
        return None, None, None, gxy, gff, None, None, None


_warned_saturation = False
_default_calls = 0


def saturated_pixels(pixel_coords, render_ranges, face_vertices_z, face_vertices_image, face_features, knum, eps=1e-8):
    """Number of pixels that MORE than knum faces cover (where NEAREST and FIRST record different faces): one extra forward with
    knum + 1 slots.  Diagnostic (bench.py reports it for BASELINE configs[4]); synchronises."""
    with torch.no_grad():
        _, face = _SparseRender.apply(pixel_coords, render_ranges, face_vertices_z, face_vertices_image, face_features,
                                      int(knum) + 1, float(eps), NEAREST)
    return int((face[..., int(knum)] >= 0).sum().item())


def deftet_sparse_render(pixel_coords, render_ranges, face_vertices_z, face_vertices_image, face_features,
                         knum=300, eps=1e-8, policy=None):
    """(pixel_coords [B,P,2], render_ranges [B,P,2] (min,max depth), face_vertices_z [B,F,3],
    face_vertices_image [B,F,3,2], face_features [B,F,3,D]) ->
    (features [B,P,knum,D] sorted nearest-first, face_idx int64 [B,P,knum], -1 = empty).
    policy: which covering faces a pixel with more than knum of them records — NEAREST (the knum nearest) or FIRST (the first knum
    in face order).  They give the same result whenever no pixel saturates, as at the reference's call site (knum = 300 against
    ~60 covering faces, deftetrneder.py:97-100); which one Kaolin implements is NOT known here (parity unpinned), so a caller
    whose knum can saturate should say which it wants: with policy=None the call renders NEAREST and — on its first call and
    every 256th after it, one small reduction + synchronisation — warns ONCE when pixels came back with all knum slots in use."""
    global _warned_saturation, _default_calls
    explicit = policy is not None
    out = _SparseRender.apply(pixel_coords, render_ranges, face_vertices_z, face_vertices_image, face_features,
                              int(knum), float(eps), int(policy) if explicit else NEAREST)
    if not explicit and not _warned_saturation:
        _default_calls += 1
        if _default_calls % 256 == 1 and not torch.cuda.is_current_stream_capturing():
            full = int((out[1][..., int(knum) - 1] >= 0).sum().item()) if knum > 0 and out[1].numel() else 0
            if full:
                import warnings
                _warned_saturation = True
                warnings.warn("deftet_sparse_render: %d pixels recorded knum=%d faces; if more faces than that cover them the saturation "
                              "policy decides which are kept (NEAREST here, the default; FIRST is the other candidate) and Kaolin's own "
                              "behaviour is not pinned in this build — pass policy=NEAREST or policy=FIRST explicitly "
                              "(deftet_amd.render.deftet_sparse_render.saturated_pixels counts the pixels that really differ)"
                              % (full, int(knum)), RuntimeWarning, stacklevel=2)
    return out
