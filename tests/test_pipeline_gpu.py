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


def _case(cuda, B=2, Q=3000, res=10):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import step_demo
    return step_demo.build_case(res, B, Q, cuda)


def test_forward_surface_align_end_to_end(cuda, oracle):
    """DefTet.forward_surface_align on the HIP-backed module, both modes, with the reference's argument list and
    tuple order (layers/DefTet/deftet.py:51-130); every piece is compared with its separately verified operator,
    and the training-mode losses back-propagate to the vertices."""
    from deftet_amd import hip_ops, surface_losses
    from deftet_amd.layers.DefTet.deftet import DefTet
    B, Q = 2, 3000
    pos0, idx, f3, t2, gt_verts, gt_faces, pts, inv_v = _case(cuda, B, Q)
    T = idx.shape[0]
    idxB = idx[None].expand(B, -1, -1).contiguous()
    m = DefTet(device=cuda)
    m.inverse_v = inv_v
    d = np.random.default_rng(0).standard_normal((B, 5000, 3))
    gt_pts = torch.from_numpy((0.3 * d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)).to(cuda)
    mesh_list = ([gt_verts[None]] * B, [gt_faces[None]] * B)          # per shape: verts [1,V,3], faces [1,F,3] like the dataloader
    pred_occ = torch.rand(B, T, device=cuda, generator=torch.Generator(device=cuda).manual_seed(2))
    kw = dict(tetrahedron_bxfx4=idxB, mesh_list=mesh_list, gt_surface_points=gt_pts, tet_face_bxfx3=f3[None].expand(B, -1, -1),
              tet_face_tet_bx4fx2=t2[None].expand(B, -1, -1))
    # ---- inference
    with torch.no_grad():
        out = m.forward_surface_align(pos0, pts, inference=True, pred_occ=pred_occ, **kw)
    assert len(out) == 10
    amips, edge, volvar, analytic, normal, center_occ, condition, boundary, pred_face, chamfer = out
    tet = hip_ops.tet_gather(pos0, idxB)
    assert np.array_equal(condition.cpu().numpy(), oracle.point_in_tet(tet.cpu().numpy(), pts.cpu().numpy()))
    assert center_occ.shape == (B, T) and 0.05 < center_occ.mean().item() < 0.25
    assert torch.equal(center_occ.bool(), hip_ops.check_sign(gt_verts[None].expand(B, -1, -1).contiguous(), gt_faces, tet.mean(2)))
    want_b = hip_ops.boundary_index(f3, t2, center_occ, mode=1)
    assert all(torch.equal(a, b) for a, b in zip(boundary, want_b)) and len(boundary) == B
    want_p = hip_ops.boundary_index(f3, t2, (pred_occ > 0.4).float(), mode=1)
    assert all(torch.equal(a, b) for a, b in zip(pred_face, want_p))
    e = hip_ops.tet_energies(tet, inv_v, pow_v=4, pow_e=4, scale=20.0)
    assert torch.equal(torch.stack([volvar, amips, edge], 1), e)
    for x in (analytic, normal, chamfer):
        assert x.shape == (1,) and torch.isfinite(x).all() and x.item() > 0
    assert m.paste_occ(pred_occ, condition.clone()).shape == (B, Q)
    # ---- training mode: 9-tuple, gradients reach the vertices through every term
    pos = pos0.clone().requires_grad_(True)
    out = m.forward_surface_align(pos, None, **kw)
    assert len(out) == 9
    amips, edge, volvar, analytic, normal, center_occ2, boundary2, chamfer, lap = out
    assert torch.equal(center_occ2, center_occ) and lap.shape == normal.shape and (lap == 0).all()
    for term in (amips.sum(), edge.sum(), analytic.sum(), normal.sum(), chamfer.sum()):
        (g,) = torch.autograd.grad(term, pos, retain_graph=True)
        assert torch.isfinite(g).all() and g.abs().sum() > 0
    # the surface terms of shape 0 alone == forward() on that shape (same random samples via the global generator)
    torch.manual_seed(7)
    a = m.forward(v_pos_bxnx3=pos0[:1], tet_bxfx4=idxB[:1], boundary_bxfx3=boundary[0][None], gt_surface_point=gt_pts[:1],
                  inverse_offset=inv_v, tet_bxfx4x3=tet[:1], calculate_amips_volume=True)
    assert len(a) == 6 and torch.equal(a[3], e[:1, 0]) and torch.equal(a[4], e[:1, 1]) and torch.equal(a[5], tet[:1])
    torch.manual_seed(7)
    c2, an2, n2 = surface_losses.surface_terms(pos0[:1], boundary[0][None], gt_pts[:1], per_face=20)
    assert torch.equal(a[0], c2) and torch.equal(a[1], an2) and torch.equal(a[2], n2)
    # empty surface -> ones (deftet.py:159-163)
    empty = m.forward(v_pos_bxnx3=pos0[:1], tet_bxfx4=idxB[:1], boundary_bxfx3=boundary[0][None][:, :0], gt_surface_point=gt_pts[:1],
                      tet_bxfx4x3=tet[:1], calculate_amips_volume=False)
    assert all(x.shape == (1,) and x.item() == 1 for x in empty)
    # laplacian_sparse vs a dense evaluation
    from deftet_amd.utils.lib.tet_point_adj.interface import Tet_point_adj
    V = pos0.shape[1]
    adj = Tet_point_adj().run(V, idx.int().cpu().numpy(), normalize=True).to(cuda)
    off = torch.randn(B, V, 3, device=cuda, generator=torch.Generator(device=cuda).manual_seed(4))
    lap = m.laplacian_sparse(off, adj)
    dense = adj.to_dense()
    want = ((torch.einsum("vw,bwk->bvk", dense, off) - off) ** 2).sum(dim=(1, 2))
    assert torch.allclose(lap, want, rtol=1e-4)
    assert torch.allclose(dense.sum(1), torch.ones(V, device=cuda), atol=1e-5)            # rows of D^-1 A sum to one


def test_overlay_reference_shaped_callers_run_on_gpu(cuda, oracle):
    """deftet_amd.overlay.install(), then code written like the reference's callers (tests/ref_shaped_callers.py:
    reference import paths, Kaolin entry points) runs on the GPU and returns what the verified operators return."""
    import sys
    import deftet_amd.overlay as overlay
    from deftet_amd import hip_ops
    B, Q = 2, 2000
    pos0, idx, f3, t2, gt_verts, gt_faces, pts, inv_v = _case(cuda, B, Q, res=8)
    saved = dict(sys.modules)
    for k in [k for k in sys.modules if k.split(".")[0] in ("layers", "utils", "kaolin", "cv2")]:
        del sys.modules[k]
    try:
        names = overlay.install(kaolin=True, deftet_module=True)
        from tests import ref_shaped_callers as C
        from layers.DefTet.deftet import DefTet                       # resolves to the HIP-backed module
        tet = hip_ops.tet_gather(pos0, idx[None].expand(B, -1, -1).contiguous())
        occ = C.occupancy_of_centroids(tet, [gt_verts[None]] * B, [gt_faces[None]] * B)
        assert torch.equal(occ, DefTet().check_tet_inside_sdfs(tet, ([gt_verts[None]] * B, [gt_faces[None]] * B)))
        pred = torch.rand(B, idx.shape[0], device=cuda, generator=torch.Generator(device=cuda).manual_seed(0))
        cond, pasted = C.query_and_paste(tet, pts, pred)
        assert np.array_equal(cond.cpu().numpy(), oracle.point_in_tet(tet.cpu().numpy(), pts.cpu().numpy()))
        assert torch.equal(pasted, torch.gather(pred, 1, cond[..., 0].clamp(min=0).long()))
        adj = C.vertex_adjacency(pos0.shape[1], idx.int().cpu().numpy()).coalesce()
        want = oracle.tet_point_adj(idx.int().cpu().numpy(), pos0.shape[1])
        got = adj.indices().t().numpy()
        assert np.array_equal(got[np.lexsort((got[:, 1], got[:, 0]))], want[np.lexsort((want[:, 1], want[:, 0]))])
        overlay.uninstall(names)
    finally:
        for k in [k for k in sys.modules if k.split(".")[0] in ("layers", "utils", "kaolin", "cv2")]:
            del sys.modules[k]
        sys.modules.update({k: v for k, v in saved.items() if k not in sys.modules})


def test_bench_self_launches_two_ranks(cuda):
    """`python bench.py --gpus 2` with no torch.distributed environment re-launches itself under torch.distributed.run
    and rank 0 prints the JSON line (the driver's command).  Single-GPU box: both ranks share cuda:0 and the collectives
    are staged over gloo (DEFTET_BENCH_TEST_SHARED_GPU=1); on a multi-GPU node the same command runs over RCCL."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DEFTET_BENCH_TEST_SHARED_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "1"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["steps"] == 3 and rec["value"] > 0
    assert rec["roofline"]["launches_timed"] == 3 and 0 < rec["roofline"]["frac"] < 1      # the warm-up's samples are dropped
    assert rec["ms_per_step_max"] >= rec["ms_per_step_median"] > 0
    assert "cpu_baseline" not in rec                         # rank 0 at N=1 only


def test_bench_eight_ranks_config3_shared_gpu(cuda):
    """`python bench.py --gpus 8 --config 3` — BASELINE configs[3] as the driver launches it on an 8-GPU node (batch 64 =
    8 shapes per rank) — through bench.py's own launcher with all eight ranks on this box's single GPU (collectives staged
    over gloo): per-rank seeding, the LossGather flush inside the timed region, the per-rank timing fields and the
    `rccl_ranks` field are exercised at N = 8."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DEFTET_BENCH_TEST_SHARED_GPU="1", OMP_NUM_THREADS="4")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--config", "3"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["rccl_ranks"] == 8 and rec["steps"] == 2 and rec["value"] > 0
    assert rec["config"]["n_tet"] == 750000 and rec["config"]["batch_per_gpu"] == 8
    by = rec["ms_per_step_by_rank"]
    assert len(by["all"]) == 8 and by["min"] <= by["max"] and abs(by["max"] - rec["ms_per_step"]) < 1e-3
    # what a first-time RCCL run relies on: inputs generated on the GPU, every rank identified in the line (a shared-GPU
    # test run has ONE distinct device; a real run asserts eight), the transport probe passed before the timed region
    assert rec["config"]["inputs_generated_on"] == "gpu"
    assert [r["rank"] for r in rec["ranks"]] == list(range(8)) and rec["distinct_devices"] == 1 and rec["backend"] == "gloo"


def test_forward_surface_align_save_writes_the_reference_obj_files(cuda, oracle, tmp_path):
    """save=True (eval.py --save): `<save_name>_device_<d>_<i>.obj` for the first five shapes, three `v` lines and one
    `f a c b` line per boundary triangle in the reference's `%f` format (layers/DefTet/deftet.py:72-80,
    utils/mesh_utils.py:258-267); and the vertex->tet topology cache is shared across module instances (what a
    DataParallel replica is)."""
    from deftet_amd.layers.DefTet import deftet as D
    B, Q = 2, 500
    pos0, idx, f3, t2, gt_verts, gt_faces, pts, inv_v = _case(cuda, B, Q)
    idxB = idx[None].expand(B, -1, -1).contiguous()
    m = D.DefTet(device=cuda)
    m.inverse_v = inv_v
    d = np.random.default_rng(0).standard_normal((B, 2000, 3))
    gt_pts = torch.from_numpy((0.3 * d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)).to(cuda)
    kw = dict(tetrahedron_bxfx4=idxB, mesh_list=([gt_verts[None]] * B, [gt_faces[None]] * B), gt_surface_points=gt_pts,
              tet_face_bxfx3=f3[None].expand(B, -1, -1), tet_face_tet_bx4fx2=t2[None].expand(B, -1, -1))
    with torch.no_grad():
        out = m.forward_surface_align(pos0, None, save=True, save_name=str(tmp_path / "shape"), **kw)
    boundary = out[6]
    for i in range(B):
        path = tmp_path / ("shape_device_%d_%d.obj" % (torch.cuda.current_device(), i))
        lines = path.read_text().splitlines()
        F = boundary[i].shape[0]
        assert len(lines) == 4 * F and F > 0
        tri = pos0[i][boundary[i].long()].cpu().numpy()
        assert lines[0] == 'v %f %f %f' % tuple(tri[0, 0]) and lines[3] == 'f 1 3 2'
        assert lines[4 * (F - 1) + 3] == 'f %d %d %d' % (3 * F - 2, 3 * F, 3 * F - 1)
    # a second module instance (a replica) with a fresh copy of the indices reuses the cached topology
    before = dict(D._TOPOLOGIES)
    m2 = D.DefTet(device=cuda)
    t2_ = m2.gather_tet_pos(pos0, idxB.clone())
    assert torch.equal(t2_, m.gather_tet_pos(pos0, idxB))
    assert set(D._TOPOLOGIES) == set(before) and all(D._TOPOLOGIES[k][0] is before[k][0] for k in before)
    # ... and different indices of the same shape replace the entry instead of hitting it
    other = idxB.clone()
    other[:, 0] = other[:, 1]
    m2.gather_tet_pos(pos0, other)
    key = (other.device, tuple(other.shape), pos0.shape[1])
    assert D._TOPOLOGIES[key][0] is not before[key][0]
    # the same tet list for every shape (what the reference passes) is kept ONCE; lists that differ keep their own copies;
    # both give torch.gather's values and its gradient
    assert before[key][0].tet_idx.shape[0] == 1
    mixed = idxB.clone()
    mixed[1] = mixed[1].flip(0)
    for ix, rows in ((idxB, 1), (mixed, B)):
        p1, p2 = pos0.clone().requires_grad_(True), pos0.clone().requires_grad_(True)
        got = m2.gather_tet_pos(p1, ix)
        assert D._TOPOLOGIES[key][0].tet_idx.shape[0] == rows
        want = torch.gather(p2, 1, ix.long().reshape(B, -1, 1).expand(-1, -1, 3)).reshape(B, -1, 4, 3)
        assert torch.equal(got, want)
        w = torch.linspace(0.5, 1.5, got.numel(), device=cuda).reshape(got.shape)
        (got * w).sum().backward()
        (want * w).sum().backward()
        assert torch.allclose(p1.grad, p2.grad, rtol=1e-5, atol=1e-6)


def test_forward_surface_align_equals_the_reference_composition(cuda):
    """The opt-in DefTet module against tests/golden/forward_composition.npz — the reference's own
    DefTet.forward_surface_align (layers/DefTet/deftet.py:51-130) and per-shape DefTet.forward (:138-184) run on the CPU
    with the L1 operators replaced by oracle calls: the order and weighting of the per-shape terms, the 20 samples per face
    (replayed with the reference's own random numbers), the means and the return tuples of both branches."""
    import os
    from deftet_amd.layers.DefTet.deftet import DefTet
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "forward_composition.npz"))
    B = g["pos"].shape[0]
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(cuda) if dt is None else torch.from_numpy(np.ascontiguousarray(a)).to(cuda).to(dt)
    pos = t(g["pos"])
    tet_b = t(g["tets"], torch.int64)[None].expand(B, -1, -1).contiguous()
    mesh_list = ([t(g["gt_verts_%d" % i])[None] for i in range(B)], [t(g["gt_faces_%d" % i], torch.int64)[None] for i in range(B)])
    m = DefTet(device=cuda)
    m.inverse_v = t(g["inverse_v"])
    common = dict(tetrahedron_bxfx4=tet_b, mesh_list=mesh_list, gt_surface_points=t(g["gt_points"]),
                  tet_face_bxfx3=t(g["face_fx3"], torch.int64)[None], tet_face_tet_bx4fx2=t(g["tetidx_fx2"], torch.int64)[None])

    def uv_of(prefix):
        fmax = max(g["train_boundary_%d" % i].shape[0] for i in range(B))
        uv = torch.zeros(2, B, fmax, 20)
        for i in range(B):
            F = g["train_boundary_%d" % i].shape[0]
            uv[0, i, :F] = torch.from_numpy(g[prefix % ("sqrt_u", i)][0, :, :, 0])
            uv[1, i, :F] = torch.from_numpy(g[prefix % ("v", i)][0, :, :, 0])
        return uv.to(cuda)

    close = lambda got, want, tol=2e-5: np.allclose(got.detach().cpu().numpy().reshape(np.shape(want)), want, rtol=tol, atol=tol * np.abs(want).max())
    # --- training branch: (amips, edge, volume variance, analytic, normal, center_occ, boundary, chamfer, lap_v_loss)
    m.sample_uv = uv_of("rand_%s_%d")
    out = m.forward_surface_align(pos, None, inference=False, **common)
    assert len(out) == 9
    for k, name in ((0, "amips_energy"), (1, "edge"), (2, "volume_variance"), (3, "sum_analytic_distance"), (4, "sum_normal_loss"),
                    (7, "sum_chamfer_distance"), (8, "lap_v_loss")):
        assert close(out[k], g["train_" + name]), (name, out[k], g["train_" + name])
        assert tuple(out[k].shape) == g["train_" + name].shape, name
    assert np.array_equal(out[5].cpu().numpy(), g["train_center_occ"])
    for i in range(B):
        assert np.array_equal(out[6][i].cpu().numpy(), g["train_boundary_%d" % i])
    # the per-shape terms the batch terms are made of (DefTet.forward, one shape at a time; fresh samples: only the two
    # sample-free terms are compared)
    for i in range(B):
        bnd = t(g["train_boundary_%d" % i], torch.int64)[None]
        ch, an, no = m.forward(v_pos_bxnx3=pos[i:i + 1], tet_bxfx4=tet_b[i:i + 1], boundary_bxfx3=bnd, gt_surface_point=common["gt_surface_points"][i:i + 1],
                               inverse_offset=m.inverse_v, calculate_amips_volume=False)
        assert close(an, g["per_shape_terms"][i, 1]) and close(no, g["per_shape_terms"][i, 2], 5e-5), (i, an, no, g["per_shape_terms"][i])
        assert abs(float(ch) - g["per_shape_terms"][i, 0]) < 0.25 * g["per_shape_terms"][i, 0]          # other samples, same surface
    # --- inference branch: (..., center_occ, condition, boundary, pred_surface_face, chamfer)
    m.sample_uv = uv_of("rand_%s_infer_%d")
    out = m.forward_surface_align(pos, t(g["queries"]), inference=True, pred_occ=t(g["pred_occ"]), **common)
    assert len(out) == 10
    for k, name in ((0, "amips_energy"), (1, "edge"), (2, "volume_variance"), (3, "sum_analytic_distance"), (4, "sum_normal_loss"),
                    (9, "sum_chamfer_distance")):
        assert close(out[k], g["infer_" + name]), (name, out[k], g["infer_" + name])
    assert np.array_equal(out[5].cpu().numpy(), g["infer_center_occ"]) and np.array_equal(out[6].cpu().numpy(), g["infer_condition"])
    for i in range(B):
        assert np.array_equal(out[7][i].cpu().numpy(), g["infer_boundary_%d" % i])
        assert np.array_equal(out[8][i].cpu().numpy(), g["infer_pred_surface_%d" % i])
