"""Differentiable point -> triangle-soup squared distance with the reference's operator surface.

Mirrors /root/reference/layers/DefTet/tet_analytic_distance_batch/utils.py:35-79:
`tet_analytic_distance_f_batch(gt_point_clouds_bxpx3, face_bxfx3x3, n_face_b)
 -> (closest_d f32 [B,P,1], closest_f f32 [B,P,1])`, gradient only to face_bxfx3x3.
Set DEFTET_HIP_DETERMINISTIC=1 for a run-to-run reproducible backward (point-ordered
per-face reduction instead of floating-point atomics).
"""
import os

import torch
from torch.autograd import Function

from deftet_amd import hip_ops


class VarianceFunc(Function):
    @staticmethod
    def forward(ctx, gt_point_clouds_bxpx3, face_bxfx3x3, n_face_b):
        face_bxfx3x3 = face_bxfx3x3.contiguous()
        gt_point_clouds_bxpx3 = gt_point_clouds_bxpx3.contiguous()
        n_face_b = n_face_b.contiguous()
        need_grad = bool(ctx.needs_input_grad[1])
        order = None
        if need_grad:
            closest_d, closest_f, order = hip_ops.tri_dist_fwd(gt_point_clouds_bxpx3, face_bxfx3x3, n_face_b, want_order=True)
        else:
            closest_d, closest_f = hip_ops.tri_dist_fwd(gt_point_clouds_bxpx3, face_bxfx3x3, n_face_b)
        ctx.has_order = order is not None
        ctx.save_for_backward(gt_point_clouds_bxpx3, face_bxfx3x3, closest_f, *([order] if order is not None else []))
        return closest_d, closest_f

    @staticmethod
    def backward(ctx, dl_dclosest_d, dl_dcloest_f):
        gt_point_clouds_bxpx3, face_bxfx3x3, closest_f = ctx.saved_tensors[:3]
        order = ctx.saved_tensors[3] if ctx.has_order else None
        det = os.environ.get("DEFTET_HIP_DETERMINISTIC", "0") not in ("", "0")
        dldtet_bxfx3x3 = hip_ops.tri_dist_bwd(gt_point_clouds_bxpx3, face_bxfx3x3, closest_f,
                                              dl_dclosest_d.contiguous(), deterministic=det, order=order)
        return None, dldtet_bxfx3x3, None


tet_analytic_distance_f_batch = VarianceFunc.apply
