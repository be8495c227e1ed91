#!/usr/bin/env python
"""Per-stage time of the surface terms of ONE shape (DefTet.forward -> surface_losses.surface_terms) at the sizes of the
geometry step demo: A8 normal consistency, sampling, A10 chamfer, A9 point-to-surface, and their backward.
    python tools/probes/surface_terms_probe.py [--res 70 --gt-points 100000]"""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import step_demo  # noqa: E402
from deftet_amd import hip_ops, surface_losses as SL  # noqa: E402
from deftet_amd.layers.DefTet.deftet import DefTet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=70)
ap.add_argument("--gt-points", type=int, default=100000)
a = ap.parse_args()
dev = torch.device("cuda:0")
pos0, idx, f3, t2, gt_verts, gt_faces, pts, inv_v = step_demo.build_case(a.res, 1, 1000, dev)
m = DefTet(device=dev)
tet = m.gather_tet_pos(pos0, idx[None])
occ = m.check_tet_inside_sdfs(tet, ([gt_verts[None]], [[gt_faces]]))
boundary = m.get_boundary_index(f3, t2, occ.squeeze(-1))[0][None]
per_face = max(1, a.gt_points // gt_faces.shape[0])
gt = SL.sample_on_faces(gt_verts[gt_faces.long()][None], per_face, torch.Generator(device=dev).manual_seed(5)).reshape(1, -1, 3).contiguous()
v = pos0.clone().requires_grad_(True)
times = {}


def timed(name, fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    times[name] = round((time.perf_counter() - t0) / reps * 1e3, 3)
    return out


tri = timed("corners", lambda: SL.corners(v, boundary))
timed("normal_consistency (A8 + torch)", lambda: SL.normal_consistency(v, boundary))
timed("A8 face_edge_adj alone", lambda: hip_ops.face_edge_adj(tri[0].detach().float(), 30))
samples = timed("sample_on_faces", lambda: SL.sample_on_faces(tri, 20).reshape(1, -1, 3))
timed("A10 nn_index alone (samples -> gt)", lambda: hip_ops.nn_index(samples.detach(), gt))
timed("cloud_to_cloud (A10 + torch)", lambda: SL.cloud_to_cloud(samples, gt))
nf = torch.full((1,), float(tri.shape[1]), device=dev)
timed("A9 tri_dist_fwd alone (gt -> surface)", lambda: hip_ops.tri_dist_fwd(gt, tri.detach().contiguous(), nf))
timed("cloud_to_surface (A9 + torch)", lambda: SL.cloud_to_surface(gt, tri))


def full():
    v.grad = None
    c, an, no = SL.surface_terms(v, boundary, gt, per_face=20)
    (c.sum() + an.sum() + no.sum()).backward()


timed("surface_terms fwd+bwd", full, reps=3)
print(json.dumps({"n_boundary_face": int(boundary.shape[1]), "n_samples": int(samples.shape[1]), "n_gt_points": int(gt.shape[1]), "ms": times}))


def bwd_of(name, build):
    def f():
        v.grad = None
        build().sum().backward()
    timed(name, f, reps=3)


times.clear()
bwd_of("normal term fwd+bwd", lambda: SL.normal_consistency(v, boundary))
bwd_of("chamfer term fwd+bwd", lambda: SL.cloud_to_cloud(SL.sample_on_faces(SL.corners(v, boundary), 20).reshape(1, -1, 3), gt).mean(-1))
bwd_of("analytic term fwd+bwd", lambda: SL.cloud_to_surface(gt, SL.corners(v, boundary)).mean(-1).mean(-1))
bwd_of("corners only fwd+bwd", lambda: SL.corners(v, boundary))
print(json.dumps({"backward_ms": times}))
