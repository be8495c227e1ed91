"""End-to-end chain of the operators as forward_surface_align composes them
(layers/DefTet/deftet.py:52-130, geometry only): every gradient reaches the vertices through the
atomic-free gather backward, and the whole is the sum of its separately verified parts."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_geometry_step_chain(cuda):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import step_demo
    from deftet_amd.layers.DefTet.deftet import DefTet
    from deftet_amd.layers.DefTet.check_condition_tetrahedron_base.utils import point_in_tet_occ
    B, Q = 2, 3000
    pos0, idx, f3, t2, gt_verts, gt_faces, pts, inv_v = step_demo.build_case(10, B, Q, cuda)
    idxB = idx[None].expand(B, -1, -1).contiguous()
    T = idx.shape[0]
    pred = torch.rand(B, T, device=cuda, generator=torch.Generator(device=cuda).manual_seed(1), requires_grad=True)
    m = DefTet(device=cuda)
    pos = pos0.clone().requires_grad_(True)
    loss, boundary, cond = step_demo.run_step(m, pos, idxB, f3, t2, gt_verts, gt_faces, pts, inv_v, pred)
    g_all, gp_all = pos.grad.clone(), pred.grad.clone()
    assert torch.isfinite(loss) and torch.isfinite(g_all).all() and g_all.abs().sum() > 0 and gp_all.abs().sum() > 0
    assert len(boundary) == B and all(b.shape[1] == 3 and b.shape[0] > 0 for b in boundary)
    # the occupancy of the centroids of the unjittered selection survives the jitter for most tets
    occ = m.check_tet_inside_sdfs(m.gather_tet_pos(pos, idxB).detach(), ([gt_verts[None]] * B, [[gt_faces]] * B))
    assert 0.05 < occ.mean().item() < 0.25
    # linearity: the same gradient from the two halves of the loss evaluated separately
    def grad_of(fn):
        p = pos0.clone().requires_grad_(True)
        tet = m.gather_tet_pos(p, idxB)
        fn(tet).backward()
        return p.grad
    def point_terms(tet):
        c, w, o = point_in_tet_occ(tet, pts, pred.detach())
        return (w * w).sum() + (o - 0.5).pow(2).sum()
    def energy_terms(tet):
        vv, am, el = m.energies(tet, inv_v)
        return 1e-3 * am.sum() + 1e-3 * el.sum() + 1e-6 * vv.sum()
    g_sum = grad_of(point_terms) + grad_of(energy_terms)
    assert (g_all - g_sum).abs().max() <= 1e-5 * g_all.abs().max()
