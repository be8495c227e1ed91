#!/usr/bin/env python
"""Generates tests/golden/forward_composition.npz by RUNNING the reference's layers/DefTet/deftet.py:51-184
(`DefTet.forward_surface_align`, training and inference branches, and the per-shape `DefTet.forward` it loops over) on the
CPU in the authoring container.  The operators the reference routes to CUDA extensions / Kaolin are replaced, for the
generation only, by functions of their ARGUMENTS computed with this repository's CPU oracle (oracle/): check_sign,
check_condition_f_base, tet_face_adj_m_f_idx, NearestNeighbor, tet_analytic_distance_f_batch.  What the fixture pins is
therefore the COMPOSITION — the order and weighting of the per-shape terms, `sample_surf_point_batch(..., 20)`, the
means, the tuple layout — not those operators (they have their own tests).  The uniform random numbers the reference
draws for its surface samples are recorded, so that a replay can use the very same sample points.

Only inputs, recorded random numbers and outputs are stored — no reference source.

    python tests/golden/gen_forward_composition.py          # needs /root/reference
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from deftet_amd import grids  # noqa: E402
from oracle import oracle as ORC  # noqa: E402


def octahedron(r):
    v = np.array([[r, 0, 0], [-r, 0, 0], [0, r, 0], [0, -r, 0], [0, 0, r], [0, 0, -r]], np.float32)
    f = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]], np.int64)
    return v, f


def main():
    # ---- stubs (functions of their arguments, computed by the CPU oracle)
    kal = types.ModuleType("kaolin")
    kal.ops = types.SimpleNamespace(mesh=types.SimpleNamespace(
        check_sign=lambda v, f, p, hash_resolution=512: torch.from_numpy(ORC.check_sign(v.numpy(), f.numpy(), p.numpy()))))
    m_cond = types.ModuleType("layers.DefTet.check_condition_tetrahedron_base.utils")
    m_cond.check_condition_f_base = lambda tet, pts: torch.from_numpy(ORC.point_in_tet(tet.detach().numpy(), pts.numpy()))
    m_adj = types.ModuleType("layers.DefTet.tet_face_adj_m_idx.utils")

    def face_pairs(face):
        tab = ORC.face_edge_adj(face.detach().numpy())
        fi, ki = np.nonzero(tab >= 0)
        return torch.from_numpy(np.stack([fi, tab[fi, ki].astype(np.int64)]))
    m_adj.tet_face_adj_m_f_idx = face_pairs
    m_nn = types.ModuleType("layers.nearest_neighbor")

    class _NN:
        def __call__(self, a, b):
            return torch.from_numpy(ORC.nn_index(a.detach().numpy(), b.detach().numpy()).astype(np.int64))
    m_nn.NearestNeighbor = _NN
    m_tri = types.ModuleType("layers.DefTet.tet_analytic_distance_batch.utils")

    def tri_dist(pts, faces, n_face):
        d, f = ORC.tri_dist_fwd(pts.detach().numpy(), faces.detach().numpy(), n_face.numpy())
        return torch.from_numpy(d), torch.from_numpy(f)
    m_tri.tet_analytic_distance_f_batch = tri_dist
    for m in (kal, m_cond, m_adj, m_nn, m_tri, types.ModuleType("cv2")):
        sys.modules[m.__name__] = m
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    # utils/tet_utils.py dlopens utils/lib/*/run.so relative to the working directory at import; the forward pass needs
    # none of them: a scratch copy with the four libraries built is what gen_golden.py prepares — reuse it
    sys.path.insert(0, HERE)
    import gen_golden
    gen_golden.prepare_reference_import()
    from layers.DefTet.deftet import DefTet
    from utils import mesh_utils as MU

    # ---- the case: res-6 Kuhn grid, three jittered shapes, an octahedron of a different size as each shape's ground truth
    res, B = 6, 3
    verts, tets = grids.kuhn_grid(res)
    pos = torch.from_numpy(grids.jittered_positions(verts, res, B, 0.1))
    f3, t2, _, _, _ = ORC.tet_to_face(tets, verts.shape[0])
    radii = (0.30, 0.36, 0.24)
    rng = np.random.default_rng(77)
    gt_v, gt_f, gt_pts = [], [], []
    for r in radii:
        v, f = octahedron(r)
        gt_v.append(v)
        gt_f.append(f)
        w = rng.dirichlet([1, 1, 1], (150,)).astype(np.float32)
        tri = v[f][rng.integers(0, 8, 150)]
        gt_pts.append((tri * w[:, :, None]).sum(1))
    gt_pts = np.stack(gt_pts).astype(np.float32)
    mesh_list = ([torch.from_numpy(v)[None] for v in gt_v], [torch.from_numpy(f)[None] for f in gt_f])
    D = DefTet()
    D.inverse_v = D.tet_inverse_v(torch.from_numpy((verts - 0.5).astype(np.float32)), torch.from_numpy(tets).long())
    tet_b = torch.from_numpy(tets).long()[None].expand(B, -1, -1).contiguous()
    pts = torch.from_numpy(grids.random_queries(B, 200, seed0=4242))
    pred_occ = torch.rand(B, tets.shape[0], generator=torch.Generator().manual_seed(9))

    # record the uniform numbers of sample_surf_point_batch (two torch.rand calls per shape: sqrt(u), v)
    draws = []
    real_rand = torch.rand

    def recording_rand(*a, **k):
        k.pop("device", None)
        out = real_rand(*a, **k)
        draws.append(out.clone())
        return out
    MU.torch = types.SimpleNamespace(**{n: getattr(torch, n) for n in dir(torch) if not n.startswith("__")})
    MU.torch.rand = recording_rand
    torch.manual_seed(2024)
    common = dict(tetrahedron_bxfx4=tet_b, mesh_list=mesh_list, gt_surface_points=torch.from_numpy(gt_pts),
                  tet_face_bxfx3=torch.from_numpy(f3).long()[None], tet_face_tet_bx4fx2=torch.from_numpy(t2).long()[None])
    train = D.forward_surface_align(pos, None, inference=False, **common)
    n_train = len(draws)
    infer = D.forward_surface_align(pos, pts, inference=True, pred_occ=pred_occ, **common)
    os.chdir(cwd)
    # per-shape terms of the training pass (deftet.py:89-110), re-run with the recorded numbers replayed
    replay = iter([d.clone() for d in draws[:n_train]])
    MU.torch.rand = lambda *a, **k: next(replay)
    boundary = train[6]
    per_shape = []
    for i in range(B):
        per_shape.append([float(x) for x in D.forward(v_pos_bxnx3=pos[i:i + 1], tet_bxfx4=tet_b[i:i + 1], boundary_bxfx3=boundary[i].unsqueeze(0),
                                                      gt_surface_point=torch.from_numpy(gt_pts[i:i + 1]), inverse_offset=D.inverse_v,
                                                      tet_bxfx4x3=None, calculate_amips_volume=False)])
    out = dict(res=res, tets=tets, verts=verts, pos=pos.numpy(), face_fx3=f3, tetidx_fx2=t2, gt_points=gt_pts, queries=pts.numpy(),
               pred_occ=pred_occ.numpy(), inverse_v=D.inverse_v.numpy(), per_shape_terms=np.array(per_shape, np.float32),
               n_draws_train=n_train)
    for i in range(B):
        out["gt_verts_%d" % i] = gt_v[i]
        out["gt_faces_%d" % i] = gt_f[i]
        out["rand_sqrt_u_%d" % i] = draws[2 * i].numpy()              # torch.rand before the sqrt (mesh_utils.py:296)
        out["rand_v_%d" % i] = draws[2 * i + 1].numpy()
        out["train_boundary_%d" % i] = train[6][i].numpy()
        out["infer_boundary_%d" % i] = infer[7][i].numpy()
        out["infer_pred_surface_%d" % i] = infer[8][i].numpy()
        out["rand_sqrt_u_infer_%d" % i] = draws[n_train + 2 * i].numpy()
        out["rand_v_infer_%d" % i] = draws[n_train + 2 * i + 1].numpy()
    names_train = ("amips_energy", "edge", "volume_variance", "sum_analytic_distance", "sum_normal_loss", "center_occ", None,
                   "sum_chamfer_distance", "lap_v_loss")
    for n, v in zip(names_train, train):
        if n:
            out["train_" + n] = v.detach().numpy()
    names_inf = ("amips_energy", "edge", "volume_variance", "sum_analytic_distance", "sum_normal_loss", "center_occ", "condition", None, None,
                 "sum_chamfer_distance")
    for n, v in zip(names_inf, infer):
        if n:
            out["infer_" + n] = v.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "forward_composition.npz"), **out)
    print("wrote forward_composition.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.startswith("train_") or k == "per_shape_terms"})


if __name__ == "__main__":
    main()
