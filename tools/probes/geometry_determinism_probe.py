"""The geometry step (forward_surface_align + occupancy query + backward, tools/step_demo.py) repeated on the same inputs: which of its
outputs are the same bits every time.  python tools/probes/geometry_determinism_probe.py [--res 40]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import step_demo  # noqa: E402
from deftet_amd import surface_losses  # noqa: E402
from deftet_amd.layers.DefTet.deftet import DefTet  # noqa: E402
from deftet_amd.layers.DefTet.check_condition_tetrahedron_base.utils import check_condition_f_base  # noqa: E402,F401

ap = argparse.ArgumentParser(); ap.add_argument("--res", type=int, default=40); ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
B, Q = 4, 20000
pos0, idx, f3, t2, gt_verts, gt_faces, pts, inv_v = step_demo.build_case(a.res, B, Q, dev)
tri = gt_verts[gt_faces.long()][None].expand(B, -1, -1, -1)
gt = surface_losses.sample_on_faces(tri, max(1, 20000 // gt_faces.shape[0]), torch.Generator(device=dev).manual_seed(5)).reshape(B, -1, 3).contiguous()
idxB = idx[None].expand(B, -1, -1).contiguous()
pred0 = torch.rand(B, idx.shape[0], device=dev)
m = DefTet(device=dev)
m.inverse_v = inv_v
ref, diff = None, {}
for it in range(a.reps):
    torch.manual_seed(11); torch.cuda.manual_seed_all(11)              # the chamfer term samples the surface: same samples every time
    pos = pos0.clone().requires_grad_(True); pred = pred0.clone().requires_grad_(True)
    tet = m.gather_tet_pos(pos, idxB)
    out = m.forward_surface_align(pos, pts, tetrahedron_bxfx4=idxB, mesh_list=([gt_verts[None]] * B, [[gt_faces]] * B), gt_surface_points=gt,
                                  tet_face_bxfx3=f3[None].expand(B, -1, -1), tet_face_tet_bx4fx2=t2[None].expand(B, -1, -1), tet_bxfx4x3=tet)
    amips, edge, vvar, analytic, normal, center_occ, boundary, chamfer, _ = out
    cond, w, occ = step_demo.point_in_tet_occ(tet, pts, pred)
    terms = dict(amips=amips, edge=edge, vvar=vvar, analytic=analytic, normal=normal, chamfer=chamfer, cond=cond, w=w, occ=occ)
    grads = {}
    for name in ("amips", "edge", "vvar", "analytic", "normal", "chamfer"):
        g, = torch.autograd.grad(terms[name].sum(), pos, retain_graph=True, allow_unused=True)
        grads["d_" + name] = g if g is not None else torch.zeros_like(pos)
    gp, gpred = torch.autograd.grad((w * w).sum() + ((occ - 0.5) ** 2).sum(), (pos, pred), retain_graph=False)
    grads["d_occupancy_pos"], grads["d_occupancy_pred"] = gp, gpred
    cur = {k: v.detach().clone().float() for k, v in {**terms, **grads}.items()}
    if ref is None:
        ref = cur
    else:
        for k in cur:
            if not torch.equal(cur[k].view(torch.int32), ref[k].view(torch.int32)):
                rel = ((cur[k] - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-30)).item()
                diff[k] = max(diff.get(k, 0.0), rel)
print("same bits in all %d runs: %s" % (a.reps, sorted(k for k in ref if k not in diff)))
print("differing (largest |difference| / max|value|): %s" % {k: "%.1e" % v for k, v in diff.items()})
