#!/usr/bin/env python
"""Per-stage time of surface_terms_batched at the geometry step demo's sizes (B shapes), forward pieces and backward."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import step_demo  # noqa: E402
from deftet_amd import hip_ops, surface_losses as SL  # noqa: E402
from deftet_amd.layers.DefTet.deftet import DefTet  # noqa: E402
from deftet_amd.layers.DefTet.tet_analytic_distance_batch.utils import tet_analytic_distance_f_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=70)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--gt-points", type=int, default=100000)
a = ap.parse_args()
dev = torch.device("cuda:0")
B = a.batch
pos0, idx, f3, t2, gt_verts, gt_faces, pts, inv_v = step_demo.build_case(a.res, B, 1000, dev)
m = DefTet(device=dev)
tet = m.gather_tet_pos(pos0, idx[None].expand(B, -1, -1).contiguous())
occ = m.check_tet_inside_sdfs(tet, ([gt_verts[None]] * B, [[gt_faces]] * B))
boundary = m.get_boundary_index(f3, t2, occ.squeeze(-1))
per_face = max(1, a.gt_points // gt_faces.shape[0])
gt = SL.sample_on_faces(gt_verts[gt_faces.long()][None].expand(B, -1, -1, -1), per_face, torch.Generator(device=dev).manual_seed(5)).reshape(B, -1, 3).contiguous()
v = pos0.clone().requires_grad_(True)
counts = [int(f.shape[0]) for f in boundary]
times = {}


def timed(name, fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    times[name] = round((time.perf_counter() - t0) / reps * 1e3, 3)
    return out


faces = torch.nn.utils.rnn.pad_sequence([f.long() for f in boundary], batch_first=True)
tri = timed("corners", lambda: SL.corners(v, faces))
trid = tri.detach().float().contiguous()
adj = timed("A8 ragged", lambda: hip_ops.face_edge_adj_ragged(trid, counts, 30))
samples = timed("sample_on_faces", lambda: SL.sample_on_faces(tri, 20).reshape(B, -1, 3))
sd = samples.detach().contiguous()
timed("A10 ragged", lambda: hip_ops.nn_index_ragged(sd, gt, [c * 20 for c in counts]))
nf = torch.tensor(counts, device=dev, dtype=torch.float32)
timed("A9 fwd (batched call)", lambda: hip_ops.tri_dist_fwd(gt, trid, nf))
d, f = hip_ops.tri_dist_fwd(gt, trid, nf)
g = torch.ones_like(d)
timed("A9 bwd (atomic)", lambda: hip_ops.tri_dist_bwd(gt, trid, f, g))
timed("surface_terms_batched fwd", lambda: SL.surface_terms_batched(v, boundary, gt, per_face=20))


def full():
    v.grad = None
    c, an, no = SL.surface_terms_batched(v, boundary, gt, per_face=20)
    (c.sum() + an.sum() + no.sum()).backward()


timed("surface_terms_batched fwd+bwd", full, reps=3)
print(json.dumps({"batch": B, "faces": counts, "n_gt_points": int(gt.shape[1]), "ms": times}))
