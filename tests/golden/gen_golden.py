#!/usr/bin/env python
"""Generates tests/golden/*.npz by IMPORTING the reference's Python code in the authoring
container (the reference tree does not exist on the GPU box).  Only inputs and outputs are
stored — no reference source.

    python tests/golden/gen_golden.py          # needs /root/reference

What is pinned:
  builders_<grid>.npz  utils/tet_utils.py Python twins: tet_to_adj_sparse (:47-92),
                       tet_to_face_adj_sparse (:155-201), tet_adj_share (:318-367),
                       tet_to_face (:208-256), tet_to_face_withtet (:259-300); the native
                       c_* front-ends (:94-95,:203-205,:371-375); render-side
                       tet_to_face_idx(with_boundary=True) (prepare_for_wz.py:49-104) and
                       utils_tetsv.tet_adj_share (utils_tetsv.py:16-75).
  cube40_hashes.npz    sha256 of the same outputs on the shipped cube_40_tet.tet.
  bary_<grid>.npz      utils/tet_utils.py:28-45 bary_centric_tet on seeded tets/points +
                       torch-autograd gradients (the A1b oracle).
  deftet_module.npz    layers/DefTet/deftet.py methods runnable on CPU with stub modules
                       for kaolin/cv2/the JIT CUDA ops: get_boundary_index (:186-195),
                       get_internal_index (:197-203), paste_occ (:132-136),
                       volume_variance (:239-263), amips_energy (:266-298),
                       edge_length (:320-338), tet_inverse_v (:300-318).
  pit_index_<case>.npz A1 index pin: utils/tet_utils.py:28-45 bary_centric_tet evaluated (fp64) for EVERY
                       (tet, query) pair of seeded cases; expected = lowest tet index whose four reference
                       weights all exceed MARGIN, `ambiguous` = queries where a tet at or below that index
                       has its smallest weight within +-MARGIN (face/edge/vertex contacts, where the fp32
                       sign tests of check_condition_tet_for.cu:105-189 may legitimately go either way).
  cube40_grid.npz      the shipped diff_render/diftet_6_subdiv/data/cube_40_tet.tet re-encoded (data fixture).
  n4_read_tetrahedron.npz  utils/dataloder_helper.py:30-69 read_tetrahedron(res=40) run on the shipped cube_40_tet.tet
                       (copied to <tmp>/quartet/meshes/cube_0.025000_tet.tet, so that the QuarTet binary is not needed):
                       sha256 of the returned vertices (after boundary snapping), tets and interior mask + a few raw rows.
  surface_glue.npz     utils/mesh_utils.py get_normal (:42-52), get_surface_normal_loss (:16-39) and point_point_distance
                       (:360-366) run on CPU with the two CUDA operators they call replaced by index tables computed
                       beforehand (edge-adjacent face pairs, nearest-neighbour indices): pins the glue arithmetic that
                       deftet_amd/surface_losses.py re-implements.
  n3_rebuilds.npz      diff_render/diftet_6_subdiv/3_model/prepare_for_wz.py: generate_edge (:184-203),
                       generate_tet_edge_idx (:223-236), generate_subdivision (:255-301, with and
                       without a split mask), generate_point_adj_idx (:134-146), delete_tet
                       (:171-180); 3_model/deftet.py tetweights2tetneighbourweights (:316-331),
                       on a Kuhn grid and on a random soup with repeated vertices.
"""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from deftet_amd import grids  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def prepare_reference_import():
    """utils/tet_utils.py dlopens utils/lib/*/run.so relative to os.getcwd() at import
    (interface.py:13,16): copy utils/ to a scratch dir, build the four libs there with the
    reference's own commands (do_all.sh), chdir, import."""
    scratch = tempfile.mkdtemp(prefix="deftet_ref_")
    shutil.copytree(os.path.join(REF, "utils"), os.path.join(scratch, "utils"))
    for lib in ("tet_adj_share", "tet_face_adj", "tet_point_adj", "colaps_v"):
        d = os.path.join(scratch, "utils", "lib", lib)
        subprocess.check_call("g++ -w -fPIC -O2 -c run.cpp -std=c++11 -fpermissive && g++ -shared -o run.so run.o",
                              shell=True, cwd=d)
    os.chdir(scratch)
    sys.path.insert(0, scratch)
    sys.path.insert(0, os.path.join(REF, "diff_render", "diftet_6_subdiv", "3_model"))
    return scratch


def coo_rows(adj):
    adj = adj.tocoo()
    o = np.lexsort((adj.col, adj.row))
    return np.stack([adj.row[o], adj.col[o]], 1).astype(np.int64), adj.data[o].astype(np.float64)


def builders_fixture(name, verts, tets, tu, pw, tsv, store_full=True):
    n_point = verts.shape[0]
    tets64 = tets.astype(np.int64)
    out = {"verts": verts.astype(np.float64), "tets": tets.astype(np.int32)}
    # --- vertex adjacency
    adj = tu.tet_to_adj_sparse(verts, [list(map(int, t)) for t in tets64], normalize=False).coalesce()
    out["point_adj_idx"] = adj.indices().numpy().T.copy()
    adjn = tu.c_tet_to_adj_sparse(verts, tets64, normalize=True).coalesce()
    out["point_adj_norm_idx"] = adjn.indices().numpy().T.copy()
    out["point_adj_norm_val"] = adjn.values().numpy().copy()
    # --- face-face adjacency (Python twin; native c_ version on the same grid)
    fa = tu.tet_to_face_adj_sparse(verts, [list(map(int, t)) for t in tets64])
    rows, vals = coo_rows(fa.tocsr())
    out["face_adj_rows"] = rows
    out["face_adj_vals"] = vals
    rows_c, vals_c = coo_rows(tu.c_tet_to_face_adj_sparse(verts, tets64))
    assert np.array_equal(rows, rows_c) and np.array_equal(vals, vals_c)
    # --- tet adjacency through shared faces (the reference raises IndexError when no face
    #     is shared — e.g. a single tet — because its row list is empty; recorded as a flag)
    try:
        share = tu.tet_adj_share(tets64, n_point)
        share_c = tu.c_tet_adj_share(tets64, n_point, torch_t=True)
        for i in range(4):
            a = share[i].coalesce().indices().numpy().T
            b = share_c[i].coalesce().indices().numpy().T
            assert np.array_equal(a, b)
            out["adj_share_%d" % i] = a.copy()
        res = tsv.tet_adj_share(tets64, n_point)
        out["adj_share_nbr_tx4"] = np.asarray(res[-1]).astype(np.int64)
        out["adj_share_raises"] = np.zeros(1, np.int64)
    except IndexError:
        out["adj_share_raises"] = np.ones(1, np.int64)
    # --- face tables
    f3, t2, tf2, b3 = tu.tet_to_face(n_point, [list(map(int, t)) for t in tets64])
    out["face_fx3"], out["face_tetidx_fx2"], out["face_tetfaceidx_fx2"], out["boundary_fx3"] = (
        f3.astype(np.int64), t2.astype(np.int64), tf2.astype(np.int64), np.asarray(b3).astype(np.int64).reshape(-1, 3))
    g3, g2, gf2 = pw.tet_to_face_idx(n_point, tets64, with_boundary=True)
    out["facewb_fx3"], out["facewb_tetidx_fx2"], out["facewb_tetfaceidx_fx2"] = g3, g2, gf2
    out["face_withtet_4tx2"] = tu.tet_to_face_withtet(verts, [list(map(int, t)) for t in tets64]).astype(np.int64)
    if store_full:
        np.savez_compressed(os.path.join(HERE, "builders_%s.npz" % name), **out)
    return {k: sha(v) for k, v in out.items()}, out

PIT_MARGIN = 1e-4


def pit_index_case(tu, torch, tet, pts, chunk=256):
    """Reference answer for the point-in-tet query from the imported bary_centric_tet
    (utils/tet_utils.py:28-45), all T x Q pairs in float64."""
    t64 = torch.from_numpy(np.asarray(tet, np.float32)).double()
    a, b, c, d = (t64[None, :, i, :] for i in range(4))
    T, Q = t64.shape[0], pts.shape[0]
    expected = np.full(Q, -1, np.int32)
    ambiguous = np.zeros(Q, bool)
    w_exp = np.zeros((Q, 4), np.float32)
    for q0 in range(0, Q, chunk):
        p = torch.from_numpy(np.asarray(pts[q0:q0 + chunk], np.float32)).double()[:, None, :]
        W = torch.stack(tu.bary_centric_tet(a, b, c, d, p), -1)            # [q,T,4]
        wmin = W.min(-1).values.numpy()
        inside = wmin > PIT_MARGIN
        touch = np.abs(wmin) <= PIT_MARGIN
        has = inside.any(1)
        first = np.where(has, inside.argmax(1), T)
        expected[q0:q0 + chunk] = np.where(has, first, -1)
        tidx = np.arange(T)[None, :]
        ambiguous[q0:q0 + chunk] = (touch & (tidx <= first[:, None])).any(1)
        # reference weights (fp32 evaluation of the same function) of the expected tet
        sel = np.where(has, first, 0)
        t32 = torch.from_numpy(np.asarray(tet, np.float32))[sel]
        p32 = torch.from_numpy(np.asarray(pts[q0:q0 + chunk], np.float32))
        w32 = torch.stack(tu.bary_centric_tet(t32[:, 0], t32[:, 1], t32[:, 2], t32[:, 3], p32), -1).numpy()
        w_exp[q0:q0 + chunk] = np.where(has[:, None], w32, 0)
    return expected, ambiguous, w_exp


def pit_index_fixtures(tu, torch):
    rng = np.random.default_rng(2024)
    cases = {}
    for res, nq, jit in ((4, 1500, 0.15), (8, 2000, 0.1), (20, 2500, 0.1)):
        tet, pts, _, _ = grids.make_case(res, nq, 1, jit)
        cases["kuhn%d" % res] = (tet[0], pts[0])
    # overlapping soup: random well-shaped tets (half of them inverted, some duplicated), queries drawn
    # inside tets, on the cloud's box and outside it
    n = 300
    ctr = rng.uniform(-0.4, 0.4, (n, 1, 3))
    soup = (ctr + rng.uniform(-0.15, 0.15, (n, 4, 3))).astype(np.float32)
    vol = grids.tet_orientation(soup[None])[0]
    soup = soup[np.abs(vol) > 1e-4]
    soup[::2, [0, 1]] = soup[::2, [1, 0]]                                     # flip orientation of every other tet
    soup = np.concatenate([soup, soup[rng.integers(0, soup.shape[0], 20)]], 0)   # exact duplicates (lowest index wins)
    w4 = rng.dirichlet([1, 1, 1, 1], 1200).astype(np.float32)
    inside_pts = (soup[rng.integers(0, soup.shape[0], 1200)] * w4[:, :, None]).sum(1)
    qs = np.concatenate([inside_pts, rng.uniform(-0.6, 0.6, (800, 3))], 0).astype(np.float32)
    cases["soup"] = (np.ascontiguousarray(soup), qs[rng.permutation(qs.shape[0])])
    # the shipped QuarTet grid at its rest positions (train_multigpu.py:65-66 shift), tets via cube40_grid.npz
    v40, t40 = grids.read_tet(os.path.join(REF, "diff_render/diftet_6_subdiv/data/cube_40_tet.tet"))
    tet40 = (v40 - 0.5).astype(np.float32)[t40]
    cases["cube40"] = (tet40, grids.random_queries(1, 1200, seed0=2040)[0])
    for name, (tet, pts) in cases.items():
        exp, amb, w = pit_index_case(tu, torch, tet, pts)
        out = dict(pts=pts, expected=exp, ambiguous=amb, w_ref_f32=w, margin=np.float64(PIT_MARGIN))
        if name != "cube40":
            out["tet"] = tet                                                  # cube40: rebuilt from cube40_grid.npz
        np.savez_compressed(os.path.join(HERE, "pit_index_%s.npz" % name), **out)
        print("pit_index_%-7s T=%6d Q=%5d hit=%5d miss=%5d ambiguous=%4d" % (
            name, tet.shape[0], pts.shape[0], int((exp >= 0).sum()), int((exp < 0).sum()), int(amb.sum())))


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; fixtures can only be generated in the authoring container")
    prepare_reference_import()
    import torch
    from utils import tet_utils as tu
    import prepare_for_wz as pw
    import utils_tetsv as tsv

    # ---------------- builders on tiny hand grids and small synthetic grids
    one = (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], float), np.array([[0, 1, 2, 3]]))
    two = (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1]], float), np.array([[0, 1, 2, 3], [1, 2, 3, 4]]))
    k2 = grids.kuhn_grid(2)
    k4 = grids.kuhn_grid(4)
    k8 = grids.kuhn_grid(8)
    rng = np.random.default_rng(7)
    k4p = (k4[0], k4[1][rng.permutation(k4[1].shape[0])][:, [2, 0, 3, 1]])      # shuffled order + relabelled locals
    for name, (v, t) in dict(one=one, two=two, kuhn2=k2, kuhn4=k4, kuhn4perm=k4p, kuhn8=k8).items():
        builders_fixture(name, np.asarray(v, float), np.asarray(t), tu, pw, tsv)
    v40, t40 = grids.read_tet(os.path.join(REF, "diff_render/diftet_6_subdiv/data/cube_40_tet.tet"))
    np.savez_compressed(os.path.join(HERE, "cube40_grid.npz"), verts=np.asarray(v40, np.float64), tets=np.asarray(t40, np.int32))
    h40, full40 = builders_fixture("cube40", v40, t40, tu, pw, tsv, store_full=False)
    np.savez_compressed(os.path.join(HERE, "cube40_hashes.npz"),
                        **{k: np.frombuffer(bytes.fromhex(v), np.uint8) for k, v in h40.items()},
                        shapes=np.array([full40[k].shape[0] for k in sorted(full40)], np.int64),
                        keys=np.array(sorted(full40)))


    # ---------------- N4: the reference's .tet reader on the shipped grid
    sys.path.insert(0, REF)
    from utils import dataloder_helper as DH
    root = tempfile.mkdtemp(prefix="deftet_n4_")
    os.makedirs(os.path.join(root, "quartet", "meshes"))
    shutil.copy(os.path.join(REF, "diff_render/diftet_6_subdiv/data/cube_40_tet.tet"),
                os.path.join(root, "quartet", "meshes", "cube_%f_tet.tet" % (1.0 / 40)))
    rv, rt, rm = DH.read_tetrahedron(res=40, root=root)
    np.savez_compressed(os.path.join(HERE, "n4_read_tetrahedron.npz"),
                        verts_sha=np.frombuffer(bytes.fromhex(sha(rv.astype(np.float64))), np.uint8),
                        tets_sha=np.frombuffer(bytes.fromhex(sha(rt.astype(np.int64))), np.uint8),
                        mask_sha=np.frombuffer(bytes.fromhex(sha(rm.astype(np.uint8))), np.uint8),
                        shape=np.array([rv.shape[0], rt.shape[0]], np.int64), n_interior=np.int64(rm.sum()),
                        n_snapped_to_0=np.int64((rv == 0).sum()), n_snapped_to_1=np.int64((rv == 1).sum()),
                        first_rows=rv[:16].astype(np.float64), res=np.float64(1.0 / 40))

    # ---------------- surface-loss glue of utils/mesh_utils.py (CUDA operators stubbed by precomputed index tables)
    from oracle import oracle as ORC
    verts16, tets16 = grids.kuhn_grid(8)
    pos16 = grids.jittered_positions(verts16, 8, 1)
    f3o, t2o, _, _, _ = ORC.tet_to_face(tets16, verts16.shape[0])
    occ16 = np.linalg.norm(pos16[0][tets16].mean(1), axis=1) < 0.3
    o2 = occ16[t2o]
    selb = o2.sum(1) == 1
    bnd = f3o[selb].copy()
    bnd[o2[selb][:, 0]] = bnd[o2[selb][:, 0]][:, ::-1]
    tri = pos16[0][bnd]                                                     # [F,3,3] float32
    tab = ORC.face_edge_adj(tri)
    fi, ki = np.nonzero(tab >= 0)
    pairs = np.stack([fi, tab[fi, ki].astype(np.int64)])
    rng_s = np.random.default_rng(21)
    src = rng_s.uniform(-0.35, 0.35, (1, 500, 3)).astype(np.float32)
    dst = rng_s.uniform(-0.35, 0.35, (1, 700, 3)).astype(np.float32)
    nn = ((src[0][:, None, :].astype(np.float64) - dst[0][None].astype(np.float64)) ** 2).sum(-1).argmin(1)
    stub_names = ("layers.DefTet.tet_face_adj_m_idx.utils", "layers.nearest_neighbor", "layers.DefTet.tet_analytic_distance_batch.utils")
    saved = {k: sys.modules.get(k) for k in stub_names}
    m1 = types.ModuleType(stub_names[0]); m1.tet_face_adj_m_f_idx = lambda face: torch.from_numpy(pairs)
    m2 = types.ModuleType(stub_names[1])

    class _NN:
        def __call__(self, a, b):
            return torch.from_numpy(nn)[None]
    m2.NearestNeighbor = _NN
    m3 = types.ModuleType(stub_names[2]); m3.tet_analytic_distance_f_batch = None
    sys.modules.update({stub_names[0]: m1, stub_names[1]: m2, stub_names[2]: m3})
    sys.modules.pop("utils.mesh_utils", None)
    sys.path.insert(0, REF)
    from utils import mesh_utils as MU
    vt = torch.from_numpy(pos16)
    ft = torch.from_numpy(bnd)[None]
    tri_t = torch.from_numpy(tri)
    normals = MU.get_normal(tri_t[:, 0], tri_t[:, 1], tri_t[:, 2])
    nloss = MU.get_surface_normal_loss(vt, ft)
    ppd = MU.point_point_distance(torch.from_numpy(src), torch.from_numpy(dst))
    np.savez_compressed(os.path.join(HERE, "surface_glue.npz"), verts=pos16, faces=bnd, pairs=pairs, normals=normals.numpy(),
                        normal_loss=nloss.numpy(), src=src, dst=dst, nn=nn, point_point_distance=ppd.numpy())
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    sys.modules.pop("utils.mesh_utils", None)

    # ---------------- A1 index pin (reference barycentrics on every pair)
    pit_index_fixtures(tu, torch)
    if os.environ.get("GEN_GOLDEN_ONLY") in ("pit", "n4", "glue"):
        return

    # ---------------- barycentric weights + autograd gradients (A1b oracle)
    for name, res in (("kuhn4", 4), ("kuhn8", 8)):
        tet, pts, _, _ = grids.make_case(res, 64, 1)
        tet = tet[0]
        T = tet.shape[0]
        g = np.random.default_rng(11)
        pick = g.integers(0, T, 256)
        w4 = g.dirichlet([1, 1, 1, 1], 256).astype(np.float32)
        w4[::4] += g.normal(0, 0.3, (64, 4)).astype(np.float32)                 # some points outside
        p = (tet[pick] * w4[:, :, None]).sum(1).astype(np.float32)
        tt = torch.from_numpy(tet[pick]).clone().requires_grad_(True)
        pp = torch.from_numpy(p).clone().requires_grad_(True)
        wa, wb, wc, wd = tu.bary_centric_tet(tt[:, 0], tt[:, 1], tt[:, 2], tt[:, 3], pp)
        w = torch.stack([wa, wb, wc, wd], -1)
        gw = torch.from_numpy(g.standard_normal((256, 4)).astype(np.float32))
        (w * gw).sum().backward()
        t64 = torch.from_numpy(tet[pick]).double().requires_grad_(True)
        p64 = torch.from_numpy(p).double().requires_grad_(True)
        w64 = torch.stack(tu.bary_centric_tet(t64[:, 0], t64[:, 1], t64[:, 2], t64[:, 3], p64), -1)
        (w64 * gw.double()).sum().backward()
        np.savez_compressed(os.path.join(HERE, "bary_%s.npz" % name), tet=tet[pick], pts=p, grad_w=gw.numpy(),
                            w_f32=w.detach().numpy(), grad_tet_f32=tt.grad.numpy(), grad_pts_f32=pp.grad.numpy(),
                            w_f64=w64.detach().numpy(), grad_tet_f64=t64.grad.numpy(), grad_pts_f64=p64.grad.numpy())

    # ---------------- layers/DefTet/deftet.py methods with stubbed third-party imports
    for modname in ("kaolin", "cv2", "layers.DefTet.check_condition_tetrahedron_base.utils",
                    "layers.DefTet.tet_face_adj_m_idx.utils", "layers.DefTet.tet_analytic_distance_batch.utils",
                    "layers.nearest_neighbor"):
        m = types.ModuleType(modname)
        for attr in ("check_condition_f_base", "tet_face_adj_m_f_idx", "tet_analytic_distance_f_batch", "NearestNeighbor"):
            setattr(m, attr, None)
        sys.modules[modname] = m
    sys.path.insert(0, REF)
    from layers.DefTet.deftet import DefTet
    D = DefTet()
    verts, tets = grids.kuhn_grid(4)
    pos = grids.jittered_positions(verts, 4, 3, 0.1)
    tet = torch.from_numpy(grids.gather_tets(pos, tets))
    T = tets.shape[0]
    f3, t2, tf2, b3 = tu.tet_to_face(verts.shape[0], [list(map(int, t)) for t in tets])
    g = torch.Generator().manual_seed(5)
    occ = (torch.rand(3, T, generator=g) > 0.5).float()
    face_t, tidx_t = torch.from_numpy(f3).long(), torch.from_numpy(t2).long()
    bnd = D.get_boundary_index(face_t, tidx_t, occ)
    inn = D.get_internal_index(face_t, tidx_t, occ)
    pred = torch.rand(3, T, generator=g)
    cond = torch.randint(-1, T, (3, 50, 1), generator=g).float()
    c2 = cond.clone()
    pasted = D.paste_occ(pred, c2)
    init_pos = torch.from_numpy((verts - 0.5).astype(np.float32))
    inv_v = D.tet_inverse_v(init_pos, torch.from_numpy(tets).long())
    tg = tet.clone().requires_grad_(True)
    vv = D.volume_variance(tg, pow=4)
    am = D.amips_energy(tg, inv_v)
    el = D.edge_length(tg, pow=4)
    am2 = D.amips_energy(tg, inv_v, square=True)                     # (:286-287; no caller of the reference passes it)
    grads = []
    for y in (vv, am, el, am2):
        (gr,) = torch.autograd.grad(y.sum(), tg, retain_graph=True)
        grads.append(gr.numpy())
    out = dict(tets=tets, verts=verts, tet_bxtx4x3=tet.numpy(), face_fx3=f3, tetidx_fx2=t2, occ=occ.numpy(),
               pred=pred.numpy(), cond=cond.numpy(), cond_after=c2.numpy(), pasted=pasted.numpy(),
               inverse_v=inv_v.numpy(), volume_variance=vv.detach().numpy(), amips=am.detach().numpy(),
               edge_length=el.detach().numpy(), g_volume_variance=grads[0], g_amips=grads[1], g_edge_length=grads[2],
               amips_square=am2.detach().numpy(), g_amips_square=grads[3])
    for i in range(3):
        out["boundary_%d" % i] = bnd[i].numpy()
        out["internal_%d" % i] = inn[i].numpy()
    np.savez_compressed(os.path.join(HERE, "deftet_module.npz"), **out)

    # ---------------- render-side rebuilds (prepare_for_wz.py, 3_model/deftet.py)
    import prepare_for_wz as W
    for modname in ("cameraop", "config", "utils_mesh"):
        m = types.ModuleType(modname)
        for attr in ("perspective", "rootdir", "savemesh", "savemeshfweights", "savemeshfweightscolor"):
            setattr(m, attr, "/tmp" if attr == "rootdir" else None)
        sys.modules[modname] = m
    sys.modules.pop("deftet", None)
    import deftet as RD                     # diff_render/diftet_6_subdiv/3_model/deftet.py

    class _Self:
        pass

    out = {}
    rng = np.random.default_rng(11)
    verts, tets = grids.kuhn_grid(6)
    soup = rng.integers(0, 40, (120, 4)).astype(np.int64)
    soup[::7, 1] = soup[::7, 0]                                  # repeated vertices inside a tet
    for name, t, P in (("grid", tets.astype(np.int64), verts.shape[0]), ("soup", soup, 45)):
        pts = rng.standard_normal((P, 3)).astype(np.float32)
        feat = rng.standard_normal((P, 5)).astype(np.float32)
        sig = rng.random(len(t)) < 0.35
        e = W.generate_edge(t)
        te = W.generate_tet_edge_idx(t, e)
        pn, fn, tn = W.generate_subdivision(t, pts, feat)
        pn2, fn2, tn2 = W.generate_subdivision(t, pts, feat, sig)
        table, adjsum = W.generate_point_adj_idx(P, t)
        w = (rng.random((len(t), 4)) * 0.02).astype(np.float32)
        w[3] = np.nan
        kept = W.delete_tet(t, w, 0.01)
        nei = rng.integers(-1, len(t), (len(t), 4)).astype(np.int64)
        holder = _Self()
        holder.tet_neighbour_idx = nei
        nw1 = RD.Deftet.tetweights2tetneighbourweights(holder, w, 1)
        nw2 = RD.Deftet.tetweights2tetneighbourweights(holder, w, 2)
        out.update({name + "_" + k: v for k, v in dict(
            tet=t, n_point=np.int64(P), pts=pts, feat=feat, sig=sig, edges=e, tet_edge=te, sub_pts=pn, sub_feat=fn, sub_tet=tn,
            sub_tet_sig=tn2, sub_pts_sig=pn2, adj_table=table, adjsum=adjsum, weights=w, kept=kept, nei=nei, nw1=nw1, nw2=nw2).items()})
    np.savez_compressed(os.path.join(HERE, "n3_rebuilds.npz"), **out)
    print("golden fixtures written to", HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("  %-28s %8d bytes" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
