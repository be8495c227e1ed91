"""GPU parity tests for A8 (face edge adjacency), A9 (point->triangle distance fwd/bwd) and
A10 (nearest neighbour): HIP path vs the CPU oracle.  Index outputs bit-exact; float outputs
bit-exact too where the accumulation order is fixed, else 1e-5 relative."""
import numpy as np
import pytest
import torch

from deftet_amd import grids

pytestmark = pytest.mark.gpu


def sphere_surface(res, batch_idx=0, r=0.3):
    """boundary triangles of the tets whose centroid lies inside a sphere."""
    from oracle import oracle as O
    verts, tets = grids.kuhn_grid(res)
    pos = grids.jittered_positions(verts, res, batch_idx + 1)[batch_idx]
    f3, t2, _, _, _ = O.tet_to_face(tets, verts.shape[0])
    cen = pos[tets].mean(1)
    occ = (np.linalg.norm(cen, axis=1) < r)
    o2 = occ[t2]
    sel = o2.sum(1) == 1
    face = f3[sel].copy()
    flip = o2[sel][:, 0]
    face[flip] = face[flip][:, ::-1]
    return pos[face].astype(np.float32)          # [F,3,3]


def test_nn_index_bit_exact(cuda, oracle):
    from deftet_amd.layers.nearest_neighbor import NearestNeighbor
    rng = np.random.default_rng(0)
    q = rng.uniform(-0.5, 0.5, (2, 3001, 3)).astype(np.float32)
    p = rng.uniform(-0.5, 0.5, (2, 1777, 3)).astype(np.float32)
    p[:, 500:520] = p[:, 100:120]                        # exact duplicates: first index wins
    q[:, :50] = p[:, 100:150]                            # zero distances
    got = NearestNeighbor()(torch.from_numpy(q).to(cuda), torch.from_numpy(p).to(cuda))
    assert got.dtype == torch.int64
    assert np.array_equal(got.cpu().numpy(), oracle.nn_index(q, p).astype(np.int64))
    from deftet_amd import hip_ops
    assert np.array_equal(hip_ops.nn_index(torch.from_numpy(q).to(cuda), torch.from_numpy(p).to(cuda), brute=True).cpu().numpy(),
                          oracle.nn_index(q, p))
    # M = 0 / 1 / not a multiple of the unroll
    for m in (1, 2, 3, 5):
        got = NearestNeighbor()(torch.from_numpy(q).to(cuda), torch.from_numpy(p[:, :m].copy()).to(cuda))
        assert np.array_equal(got.cpu().numpy(), oracle.nn_index(q, p[:, :m]).astype(np.int64))
    with pytest.raises(NotImplementedError):
        from deftet_amd.layers.nearest_neighbor.nearest_neighbor import NearestNeighborFunction
        NearestNeighborFunction.backward(None, None)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_nn_index_grid_adversarial(cuda, oracle, seed):
    """the grid search must equal the ascending scan on clustered / degenerate / far-away inputs"""
    from deftet_amd import hip_ops
    rng = np.random.default_rng(seed)
    M, N = 4000, 2500
    p = np.concatenate([rng.normal(0, 0.01, (M // 2, 3)), rng.uniform(-1, 1, (M // 4, 3)), rng.uniform(50, 60, (M // 4, 3))]).astype(np.float32)
    if seed == 1:
        p[:, 2] = 0.25                                   # all points in one plane (flat axis)
    if seed == 2:
        p[:] = p[0]                                      # all points identical
        p[7] = p[0] + np.float32(1e-3)
    p[100:120] = p[200:220]                              # exact duplicates
    p[300] = np.nan
    p[301, 1] = np.inf
    p[302] = 3e18                                        # d ~ 2.7e37 > 1e20: never selected
    p[303] = 1e9                                         # d ~ 3e18 < 1e20: selectable
    q = np.concatenate([rng.normal(0, 0.02, (N // 2, 3)), rng.uniform(-2, 2, (N // 4, 3)), rng.uniform(-100, 100, (N // 4, 3))]).astype(np.float32)
    q[:50] = p[rng.integers(0, M, 50)]                   # zero distance (NaN rows give NaN queries too)
    q[60] = np.nan
    q[61, 0] = np.inf
    q[62] = 5e8
    q[63] = 1e12                                         # every finite point is farther than sqrt(1e20)
    for b in (p[None], p[None][:, ::-1].copy()):
        want = oracle.nn_index(q[None], b)
        got = hip_ops.nn_index(torch.from_numpy(q[None]).to(cuda), torch.from_numpy(b).to(cuda)).cpu().numpy()
        assert np.array_equal(got, want)


def test_nn_index_far_queries_full_size(cuda):
    """80,640 queries spread over the cube against 100,000 points on a sphere: most queries take the far path
    (sorted by cell, rows split over blocks, answers combined with a 64-bit atomicMin) and must equal the plain scan;
    also with duplicated points (ties resolved towards the lower index across blocks)."""
    from deftet_amd import hip_ops
    rng = np.random.default_rng(11)
    d = rng.normal(size=(100000, 3))
    gt = (0.3 * d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    gt[50000:60000] = gt[:10000]                                     # exact duplicates far apart in index
    p = torch.from_numpy(gt).to(cuda)[None]
    q = torch.from_numpy(rng.uniform(-0.3, 0.3, (1, 80640, 3)).astype(np.float32)).to(cuda)
    q[0, :64] = 0.0                                                  # one group needs every point
    assert torch.equal(hip_ops.nn_index(q, p), hip_ops.nn_index(q, p, brute=True))
    far_out = torch.from_numpy(rng.uniform(-5, 5, (1, 20000, 3)).astype(np.float32)).to(cuda)   # mostly outside the grid
    assert torch.equal(hip_ops.nn_index(far_out, p), hip_ops.nn_index(far_out, p, brute=True))


def test_nn_index_properties_full_size(cuda):
    """100k GT points (dataloader.py:169) x 60k queries: idempotence + optimality property."""
    g = torch.Generator(device=cuda).manual_seed(1)
    p = torch.rand(1, 100000, 3, device=cuda, generator=g) - 0.5
    q = torch.rand(1, 60000, 3, device=cuda, generator=g) - 0.5
    from deftet_amd import hip_ops
    idx = hip_ops.nn_index(q, p).long()
    assert torch.equal(idx, hip_ops.nn_index(q, p, brute=True).long())       # grid search == exhaustive scan, 6e9 pairs
    near = torch.gather(p, 1, idx[..., None].expand(-1, -1, 3))
    d = ((near - q) ** 2).sum(-1)
    # no sampled point is closer
    samp = p[:, ::97]
    d2 = ((q[:, :, None, :] - samp[:, None, :, :]) ** 2).sum(-1).min(-1).values
    assert (d <= d2 * (1 + 1e-5) + 1e-12).all()
    # querying the points themselves returns themselves (unique random points)
    idx2 = hip_ops.nn_index(p[:, :5000].contiguous(), p)
    assert torch.equal(idx2[0].long(), torch.arange(5000, device=cuda))


@pytest.mark.parametrize("res", [12, 20])
def test_face_edge_adj_bit_exact(cuda, oracle, res):
    from deftet_amd import hip_ops
    from deftet_amd.layers.DefTet.tet_face_adj_m_idx.utils import tet_face_adj_m_f_idx
    face = sphere_surface(res)
    F = face.shape[0]
    assert F > 100
    want = oracle.face_edge_adj(face, 30)
    got = hip_ops.face_edge_adj(torch.from_numpy(face).to(cuda), 30).cpu().numpy()
    assert np.array_equal(got, want)
    assert np.array_equal(hip_ops.face_edge_adj(torch.from_numpy(face).to(cuda), 30, brute=True).cpu().numpy(), want)
    assert ((want >= 0).sum(1) == 3).mean() > 0.9          # closed manifold surface: 3 edge neighbours
    idx = tet_face_adj_m_f_idx(torch.from_numpy(face).to(cuda))
    rows, cols = np.nonzero(want >= 0)
    assert idx.dtype == torch.int64 and idx.shape[0] == 2
    assert np.array_equal(idx.cpu().numpy(), np.stack([rows, want[rows, cols].astype(np.int64)]))
    # saturation at n_max_nei and degenerate inputs
    rep = np.repeat(face[:3], 20, axis=0)                  # 60 faces, each shares edges with 39+ others
    w2 = oracle.face_edge_adj(rep, 30)
    g2 = hip_ops.face_edge_adj(torch.from_numpy(rep).to(cuda), 30).cpu().numpy()
    assert np.array_equal(g2, w2) and (w2 >= 0).all()
    for sc in (1e-9, 3e-8, 1e-6):                          # coordinates around / below the 1e-15 L1 tolerance scale
        tiny = (face[:60] * sc).astype(np.float32)
        assert np.array_equal(hip_ops.face_edge_adj(torch.from_numpy(tiny).to(cuda), 30).cpu().numpy(), oracle.face_edge_adj(tiny, 30))
    # mixed: some coordinates tiny, zeros of both signs, non-finite vertices, a perturbed duplicate
    rng = np.random.default_rng(1)
    mix = face[:80].copy()
    mix[:, :, 2] = 0.0
    mix[::3, :, 2] = -0.0
    mix[5, 1, 0] = np.nan
    mix[6, 2, 1] = np.inf
    mix[7] = mix[8]
    mix[9] = mix[10] + np.float32(1e-16)
    mix[11, :, 1] *= np.float32(1e-9)
    assert np.array_equal(hip_ops.face_edge_adj(torch.from_numpy(mix).to(cuda), 30).cpu().numpy(), oracle.face_edge_adj(mix, 30))
    assert np.array_equal(hip_ops.face_edge_adj(torch.from_numpy(mix).to(cuda), 7).cpu().numpy(), oracle.face_edge_adj(mix, 7))
    empty = tet_face_adj_m_f_idx(torch.zeros(0, 3, 3, device=cuda))
    assert empty.numel() == 0 and empty.is_floating_point()


def _tri_case(seed, P=3000, res=16):
    rng = np.random.default_rng(seed)
    face = sphere_surface(res)
    F = face.shape[0]
    d = rng.standard_normal((P, 3))
    pts = (0.3 * d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(0.7, 1.3, (P, 1))).astype(np.float32)
    pts[:100] = face[rng.integers(0, F, 100), rng.integers(0, 3, 100)]           # on vertices
    w = rng.dirichlet([1, 1, 1], 100).astype(np.float32)
    pts[100:200] = (face[rng.integers(0, F, 100)] * w[:, :, None]).sum(1)          # on faces
    ew = rng.random((100, 1)).astype(np.float32)
    tri = face[rng.integers(0, F, 100)]
    pts[200:300] = tri[:, 0] * ew + tri[:, 1] * (1 - ew)                           # on edges
    face2 = np.concatenate([face, face[:5] * 0, np.repeat(face[5:6, :1], 3, axis=1)], 0)   # degenerate triangles
    return pts[None], face2[None].astype(np.float32)


@pytest.mark.parametrize("seed", [0, 1])
def test_tri_dist_forward_bit_exact(cuda, oracle, seed):
    from deftet_amd import hip_ops
    pts, face = _tri_case(seed)
    nfb = np.array([face.shape[1]], np.float32)
    wd, wf = oracle.tri_dist_fwd(pts, face, nfb)
    for brute in (False, True):                                      # grid search and streaming scan
        d, f = hip_ops.tri_dist_fwd(torch.from_numpy(pts).to(cuda), torch.from_numpy(face).to(cuda), torch.from_numpy(nfb).to(cuda),
                                    brute=brute)
        assert np.array_equal(f.cpu().numpy(), wf)                   # argmin face: bit-exact
        assert np.array_equal(d.cpu().numpy(), wd)                   # same op order, no FMA: bit-exact
    # ragged batch: n_face_b limits the scan
    nfb2 = np.array([face.shape[1] // 3], np.float32)
    wd2, wf2 = oracle.tri_dist_fwd(pts, face, nfb2)
    d2, f2 = hip_ops.tri_dist_fwd(torch.from_numpy(pts).to(cuda), torch.from_numpy(face).to(cuda), torch.from_numpy(nfb2).to(cuda))
    assert np.array_equal(f2.cpu().numpy(), wf2) and np.array_equal(d2.cpu().numpy(), wd2)
    assert (wf2 < nfb2[0]).all()
    # zero faces: distance stays 10000, index -1 (for.cu:277-278)
    d0, f0 = hip_ops.tri_dist_fwd(torch.from_numpy(pts).to(cuda), torch.from_numpy(face).to(cuda), torch.zeros(1, device=cuda))
    assert (d0 == 10000).all() and (f0 == -1).all()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_tri_dist_grid_adversarial(cuda, oracle, seed):
    """grid search == ascending scan on a triangle soup with vertical / sliver / tiny / huge /
    non-finite / duplicated faces and points near, on, far from and outside everything"""
    from deftet_amd import hip_ops
    rng = np.random.default_rng(seed)
    F, P = 900, 3000
    cen = rng.uniform(-0.5, 0.5, (F, 1, 3))
    face = (cen + rng.normal(0, 0.03, (F, 3, 3))).astype(np.float32)
    face[:100, :, 2] = face[:100, :1, 2] + rng.normal(0, 1e-4, (100, 3)).astype(np.float32)     # nearly horizontal
    face[100:200, :, 0] = face[100:200, :1, 0]                                                    # exactly vertical (k3 == 0)
    face[200:260, :, 1] = face[200:260, :1, 1] + rng.normal(0, 1e-4, (60, 3)).astype(np.float32) # nearly vertical
    face[260:300, 2] = face[260:300, 0] * 0.5 + face[260:300, 1] * 0.5 + np.float32(1e-6)        # slivers
    face[300:310] = (face[300:310] - cen[300:310]) * 1e-4 + cen[300:310]                          # tiny faces (|n| < 1e-5)
    face[310:313] = cen[310:313] + rng.normal(0, 2.0, (3, 3, 3))                                  # huge faces (wide list)
    face[313, 0, 0] = np.nan
    face[314, 1] = np.inf
    face[315] = face[316]                                                                         # duplicates: lower index wins
    face[317] = 0.0                                                                               # collapsed at the origin
    if seed == 2:
        face[:, :, 2] *= 1e-3                                                                     # almost planar soup (thin bbox)
    pts = rng.uniform(-0.7, 0.7, (P, 3)).astype(np.float32)
    w = rng.dirichlet([1, 1, 1], 300).astype(np.float32)
    pts[:300] = (face[rng.integers(320, F, 300)] * w[:, :, None]).sum(1)                          # on faces
    pts[300:400] = face[rng.integers(320, F, 100), rng.integers(0, 3, 100)]                       # on vertices
    pts[400:500] = rng.uniform(-30, 30, (100, 3))                                                 # far away
    pts[500] = np.nan
    pts[501, 2] = np.inf
    pts[502] = 3e6
    pts[503] = 0.0
    face, pts = face[None].astype(np.float32), pts[None].astype(np.float32)
    for nf in (F, 500, 1):
        nfb = np.array([nf], np.float32)
        wd, wf = oracle.tri_dist_fwd(pts, face, nfb)
        d, f = hip_ops.tri_dist_fwd(torch.from_numpy(pts).to(cuda), torch.from_numpy(face).to(cuda), torch.from_numpy(nfb).to(cuda))
        assert np.array_equal(f.cpu().numpy(), wf)
        assert np.array_equal(d.cpu().numpy(), wd, equal_nan=True)


def test_tri_dist_grid_equals_scan_full_size(cuda):
    """100k GT points x the res-70 sphere surface: the grid search must reproduce the streaming scan"""
    from deftet_amd import hip_ops
    face = torch.from_numpy(sphere_surface(70)).to(cuda)[None]
    d = np.random.default_rng(3000).standard_normal((100000, 3))
    gt = torch.from_numpy((0.3 * d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)).to(cuda)[None]
    nfb = torch.tensor([float(face.shape[1])], device=cuda)
    a = hip_ops.tri_dist_fwd(gt, face, nfb)
    b = hip_ops.tri_dist_fwd(gt, face, nfb, brute=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert a[0].max().item() < 1e-3 and (a[1] >= 0).all()


@pytest.mark.parametrize("scale", [1.3, 0.5, 2.5])
def test_tri_dist_far_points_full_size(cuda, scale):
    """points 30 % outside / halfway inside / far outside the surface (early training): nothing is settled by the two
    shells, every point takes the far path (cell-ordered groups, certified row pruning, LDS bitset, 64-bit atomicMin
    across blocks); must equal the streaming scan, also with duplicated faces (ties towards the lower index), two
    shapes with different face counts and a few NaN / huge points"""
    from deftet_amd import hip_ops
    f0 = sphere_surface(70)
    f1 = np.concatenate([f0[:1500], f0[:1500]], 0)                   # second shape: every face twice
    F = max(f0.shape[0], f1.shape[0])
    face = np.zeros((2, F, 3, 3), np.float32)
    face[0, :f0.shape[0]] = f0
    face[1, :f1.shape[0]] = f1
    rng = np.random.default_rng(int(scale * 10))
    d = rng.standard_normal((2, 30000, 3))
    pts = (0.3 * scale * d / np.linalg.norm(d, axis=2, keepdims=True)).astype(np.float32)
    pts[0, 5] = np.nan
    pts[1, 7] = 3e7
    pts[1, 100:164] = 0.0                                            # one group at the centre: every face within reach
    nfb = torch.tensor([float(f0.shape[0]), float(f1.shape[0])], device=cuda)
    tp, tf = torch.from_numpy(pts).to(cuda), torch.from_numpy(face).to(cuda)
    a = hip_ops.tri_dist_fwd(tp, tf, nfb)
    b = hip_ops.tri_dist_fwd(tp, tf, nfb, brute=True)
    assert torch.equal(a[1], b[1])
    assert torch.equal(a[0], b[0])
    assert (a[1][1] < 1500).all()                                    # duplicates: the lower index wins


@pytest.mark.parametrize("seed", [0, 1])
def test_tri_dist_backward(cuda, oracle, seed, monkeypatch):
    from deftet_amd import hip_ops
    from deftet_amd.layers.DefTet.tet_analytic_distance_batch.utils import tet_analytic_distance_f_batch
    pts, face = _tri_case(seed)
    nfb = np.array([face.shape[1]], np.float32)
    wd, wf = oracle.tri_dist_fwd(pts, face, nfb)
    g = np.random.default_rng(9).standard_normal(wd.shape).astype(np.float32)
    want = oracle.tri_dist_bwd(pts, face, wf, g)
    tp, tf_, tn = torch.from_numpy(pts).to(cuda), torch.from_numpy(face).to(cuda), torch.from_numpy(nfb).to(cuda)
    # deterministic mode reproduces the oracle's serial accumulation order bit for bit
    det = hip_ops.tri_dist_bwd(tp, tf_, torch.from_numpy(wf).to(cuda), torch.from_numpy(g).to(cuda), deterministic=True)
    assert np.array_equal(det.cpu().numpy(), want)
    # default (atomic) mode: 1e-5 relative to the tensor scale
    at = hip_ops.tri_dist_bwd(tp, tf_, torch.from_numpy(wf).to(cuda), torch.from_numpy(g).to(cuda))
    assert np.abs(at.cpu().numpy() - want).max() <= 1e-5 * np.abs(want).max()
    # through autograd, reference signature
    tf_g = tf_.clone().requires_grad_(True)
    d, f = tet_analytic_distance_f_batch(tp, tf_g, tn)
    (d * torch.from_numpy(g).to(cuda)).sum().backward()
    assert np.abs(tf_g.grad.cpu().numpy() - want).max() <= 1e-5 * np.abs(want).max()
    # all three closest-feature cases are exercised
    assert (want != 0).any()
    # the edge case writes only the first endpoint (back.cu:309-315): analytic gradient differs
    # from finite differences there by design — pinned against the oracle above, not against FD


def test_surface_terms_compose_like_deftet_forward(cuda, oracle):
    """normal consistency / sample->cloud / cloud->surface terms composed as DefTet.forward does
    (layers/DefTet/deftet.py:168-181), with gradients to the vertices; each term against a dense torch
    evaluation of its definition."""
    from deftet_amd import surface_losses as SL
    verts, tets = grids.kuhn_grid(16)
    pos = grids.jittered_positions(verts, 16, 1)
    f3, t2, _, _, _ = oracle.tet_to_face(tets, verts.shape[0])
    occ = (np.linalg.norm(pos[0][tets].mean(1), axis=1) < 0.3)
    o2 = occ[t2]
    sel = o2.sum(1) == 1
    bnd = f3[sel].copy()
    bnd[o2[sel][:, 0]] = bnd[o2[sel][:, 0]][:, ::-1]
    v = torch.from_numpy(pos).to(cuda).requires_grad_(True)
    faces = torch.from_numpy(bnd)[None].to(cuda)
    d = np.random.default_rng(0).standard_normal((20000, 3))
    gt = torch.from_numpy((0.3 * d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32))[None].to(cuda)
    gen = torch.Generator(device=cuda).manual_seed(0)
    chamfer, analytic, normal_loss = SL.surface_terms(v, faces, gt, per_face=20, generator=gen)
    (normal_loss.sum() + chamfer.sum() + analytic.sum()).backward()
    assert torch.isfinite(v.grad).all() and v.grad.abs().sum() > 0
    assert 0 <= normal_loss.item() < 1 and 0 < chamfer.item() < 0.1 and 0 < analytic.item() < 0.1
    tri = SL.corners(v.detach(), faces)
    assert torch.equal(tri[0], v.detach()[0][faces[0]])
    # samples lie on their triangles: barycentric reconstruction error ~ 0, and they are area-uniform on average
    smp = SL.sample_on_faces(tri, 64, generator=gen)
    cen = smp.mean(2)
    assert (cen - tri.mean(2)).abs().max() < 0.02
    # sample -> cloud against a dense torch evaluation
    pred_pts = SL.sample_on_faces(tri, 20, generator=gen).reshape(1, -1, 3)[:, :2000].contiguous()
    dd = torch.cdist(pred_pts[0], gt[0]).min(-1).values
    assert torch.allclose(SL.cloud_to_cloud(pred_pts, gt)[0], torch.sqrt(dd ** 2 + 1e-10), rtol=1e-3, atol=1e-5)
    # normal consistency against the oracle's adjacency + fp64 normals
    tab = oracle.face_edge_adj(tri[0].cpu().numpy())                   # [F,30] neighbour table, -1 padded
    fi, ki = np.nonzero(tab >= 0)
    adj = np.stack([fi, tab[fi, ki].astype(np.int64)])
    n64 = torch.linalg.cross(tri[0, :, 1].double() - tri[0, :, 0].double(), tri[0, :, 2].double() - tri[0, :, 0].double())
    n64 = n64 / torch.sqrt((n64 * n64).sum(-1, keepdim=True) + 1e-12)
    a = torch.from_numpy(np.asarray(adj)).to(cuda).long()
    want = (1 - (n64[a[0]] * n64[a[1]]).sum(-1)).mean()
    assert abs(want.item() - normal_loss.item()) < 1e-5


def test_surface_terms_vs_reference_mesh_utils_outputs(cuda):
    """normal_consistency and cloud_to_cloud on the GPU == what the reference's get_surface_normal_loss / point_point_distance
    (utils/mesh_utils.py:16-39, :360-366) returned on CPU for the same inputs (tests/golden/surface_glue.npz; there the two
    CUDA operators were replaced by precomputed index tables, which the HIP operators must reproduce)."""
    import os
    from deftet_amd import hip_ops, surface_losses as SL
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "surface_glue.npz"))
    v = torch.from_numpy(g["verts"]).to(cuda)
    f = torch.from_numpy(g["faces"])[None].to(cuda)
    tri = SL.corners(v, f)
    from deftet_amd.layers.DefTet.tet_face_adj_m_idx.utils import tet_face_adj_m_f_idx
    pairs = tet_face_adj_m_f_idx(tri[0].contiguous())
    assert np.array_equal(pairs.cpu().numpy(), g["pairs"])                      # A8 == the oracle's table the fixture was made with
    loss = SL.normal_consistency(v, f)
    assert np.abs(loss.cpu().numpy() - g["normal_loss"]).max() <= 1e-6
    src, dst = torch.from_numpy(g["src"]).to(cuda), torch.from_numpy(g["dst"]).to(cuda)
    assert np.array_equal(hip_ops.nn_index(src, dst)[0].cpu().numpy(), g["nn"])   # A10 == fp64 argmin (no ties in this cloud)
    d = SL.cloud_to_cloud(src, dst)
    assert np.abs(d.cpu().numpy() - g["point_point_distance"]).max() <= 1e-6


def _sphere_surfaces(cuda, oracle, radii, res=16):
    """Boundary face lists (different counts) of the tets inside spheres of the given radii, on one jittered grid."""
    verts, tets = grids.kuhn_grid(res)
    pos = grids.jittered_positions(verts, res, len(radii))
    f3, t2, _, _, _ = oracle.tet_to_face(tets, verts.shape[0])
    out = []
    for b, r in enumerate(radii):
        occ = np.linalg.norm(pos[b][tets].mean(1), axis=1) < r
        o2 = occ[t2]
        sel = o2.sum(1) == 1
        bnd = f3[sel].copy()
        bnd[o2[sel][:, 0]] = bnd[o2[sel][:, 0]][:, ::-1]
        out.append(torch.from_numpy(bnd.astype(np.int64)).to(cuda))
    return torch.from_numpy(pos).to(cuda), out


def test_ragged_surface_operators_equal_per_shape_calls(cuda, oracle):
    from deftet_amd import hip_ops
    """deftet_face_edge_adj_ragged_f32 / deftet_nn_index_ragged_f32 (one call, shapes on the library's shape streams) ==
    the single-shape operators shape by shape, bit for bit; an empty surface in the batch is tolerated; the batched A9
    and A10 calls (now shape-parallel inside) == their per-shape calls."""
    from deftet_amd import surface_losses as SL
    v, faces = _sphere_surfaces(cuda, oracle, [0.3, 0.0, 0.22, 0.38])
    counts = [int(f.shape[0]) for f in faces]
    assert counts[1] == 0 and len(set(counts)) == 4
    B, fmax = len(faces), max(counts)
    pad = torch.nn.utils.rnn.pad_sequence(faces, batch_first=True)
    tri = SL.corners(v, pad).contiguous()
    adj = hip_ops.face_edge_adj_ragged(tri, counts, 30)
    assert adj.shape == (B, fmax, 30)
    for b in range(B):
        want = hip_ops.face_edge_adj(tri[b, :counts[b]].contiguous(), 30) if counts[b] else adj.new_zeros(0, 30)
        assert torch.equal(adj[b, :counts[b]], want), b
        assert (adj[b, counts[b]:] == -1).all()
    # A10 ragged
    g = torch.Generator(device=cuda).manual_seed(3)
    q = (torch.rand(B, fmax * 5, 3, device=cuda, generator=g) - 0.5)
    pts = (torch.rand(B, 30000, 3, device=cuda, generator=g) - 0.5) * 0.7
    nq = [c * 5 for c in counts]
    idx = hip_ops.nn_index_ragged(q, pts, nq)
    full = hip_ops.nn_index(q, pts)
    brute = hip_ops.nn_index(q, pts, brute=True)
    assert torch.equal(full, brute)
    for b in range(B):
        assert torch.equal(idx[b, :nq[b]], full[b, :nq[b]]), b
        assert (idx[b, nq[b]:] == 0).all()
    # A9 batched (shape-parallel inside) == per-shape == streaming scan
    nf = torch.tensor(counts, device=cuda, dtype=torch.float32)
    d, f = hip_ops.tri_dist_fwd(pts[:, :8000].contiguous(), tri, nf)
    db, fb = hip_ops.tri_dist_fwd(pts[:, :8000].contiguous(), tri, nf, brute=True)
    assert torch.equal(d, db) and torch.equal(f, fb)


def test_surface_terms_batched_equal_per_shape_terms(cuda, oracle):
    """surface_terms_batched (one ragged launch sequence for the batch) == surface_terms shape by shape: the normal
    and point-to-surface terms to 1e-5 (no randomness), the chamfer term statistically (other random samples), gradients
    to the vertices close; an empty surface yields (1, 1, 1) like DefTet.forward."""
    from deftet_amd import surface_losses as SL
    v0, faces = _sphere_surfaces(cuda, oracle, [0.3, 0.0, 0.22, 0.38])
    B = len(faces)
    d = np.random.default_rng(1).standard_normal((B, 20000, 3))
    gt = torch.from_numpy((0.3 * d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)).to(cuda)
    v = v0.clone().requires_grad_(True)
    ch, an, no = SL.surface_terms_batched(v, faces, gt, per_face=20, generator=torch.Generator(device=cuda).manual_seed(0))
    (ch.sum() + an.sum() + no.sum()).backward()
    g_batched = v.grad.clone()
    assert ch[1] == 1 and an[1] == 1 and no[1] == 1
    g_ref = torch.zeros_like(g_batched)
    for b in range(B):
        if faces[b].shape[0] == 0:
            continue
        vb = v0[b:b + 1].clone().requires_grad_(True)
        c1, a1, n1 = SL.surface_terms(vb, faces[b][None], gt[b:b + 1], per_face=20, generator=torch.Generator(device=cuda).manual_seed(7))
        (c1.sum() + a1.sum() + n1.sum()).backward()
        g_ref[b] = vb.grad[0]
        assert abs(a1.item() - an[b].item()) <= 1e-5 * max(1.0, abs(a1.item())), (b, a1.item(), an[b].item())
        assert abs(n1.item() - no[b].item()) <= 1e-5, (b, n1.item(), no[b].item())
        assert abs(c1.item() - ch[b].item()) <= 0.03 * c1.item(), (b, c1.item(), ch[b].item())
    assert torch.isfinite(g_batched).all()
    assert (g_batched[1] == 0).all()
    # gradients: the deterministic terms dominate; the sampled term differs by sampling noise only
    rel = (g_batched - g_ref).norm() / g_ref.norm()
    assert rel < 0.2, rel


def test_normal_consistency_op_vs_torch_autograd(cuda, oracle):
    """The fused normal-consistency operator (forward value and gradient w.r.t. the corners) == the fp64 torch composition
    of utils/mesh_utils.py:28-39 on the same A8 table, for a ragged batch with an empty surface."""
    from deftet_amd import hip_ops, surface_losses as SL
    v, faces = _sphere_surfaces(cuda, oracle, [0.3, 0.0, 0.22])
    counts = [int(f.shape[0]) for f in faces]
    pad = torch.nn.utils.rnn.pad_sequence(faces, batch_first=True)
    tri = SL.corners(v, pad).contiguous().requires_grad_(True)
    adj = hip_ops.face_edge_adj_ragged(tri.detach(), counts, 30)
    n_face = torch.tensor(counts, device=cuda, dtype=torch.int32)
    loss = hip_ops.normal_consistency(tri, adj, n_face)
    w = torch.tensor([0.7, 1.3, -0.4], device=cuda)
    (loss * w).sum().backward()
    t64 = tri.detach().double().requires_grad_(True)
    c = torch.linalg.cross(t64[:, :, 1] - t64[:, :, 0], t64[:, :, 2] - t64[:, :, 0], dim=-1)
    n = c / torch.sqrt((c * c).sum(-1, keepdim=True) + 1e-12)
    ok = adj >= 0
    nj = torch.gather(n, 1, adj.clamp(min=0).long().reshape(len(counts), -1, 1).expand(-1, -1, 3)).reshape(len(counts), -1, 30, 3)
    want = ((1 - (n[:, :, None] * nj).sum(-1)) * ok).sum((1, 2)) / ok.sum((1, 2)).clamp(min=1)
    (want * w.double()).sum().backward()
    assert loss[1] == 0 and torch.allclose(loss.double(), want, rtol=1e-5, atol=1e-7)
    assert (tri.grad[1] == 0).all()
    assert torch.allclose(tri.grad.double(), t64.grad, rtol=2e-4, atol=1e-6 * t64.grad.abs().max().item())


@pytest.mark.parametrize("shape", [(8, 97344, 1), (3, 1001), (1, 7, 5), (5, 0)])
def test_sqrt_rowsum_vs_torch_autograd(cuda, shape):
    """hip_ops.sqrt_rowsum (the tail of the point-to-surface term: sqrt(d^2 + 1e-10) summed per shape) == the fp64 torch
    composition, value and gradient; aligned and unaligned row lengths, an empty row set."""
    from deftet_amd import hip_ops
    g = torch.Generator(device=cuda).manual_seed(11)
    x = (torch.rand(*shape, device=cuda, generator=g) ** 4).requires_grad_(True)          # many values near zero, where the eps matters
    w = torch.linspace(-1.0, 2.0, shape[0], device=cuda)
    out = hip_ops.sqrt_rowsum(x, 1e-10)
    (out * w).sum().backward()
    x64 = x.detach().double().requires_grad_(True)
    want = torch.sqrt(x64 + 1e-10).reshape(shape[0], -1).sum(-1)
    (want * w.double()).sum().backward()
    assert out.shape == (shape[0],) and torch.allclose(out.double(), want, rtol=1e-5, atol=1e-7)
    assert torch.allclose(x.grad.double(), x64.grad, rtol=1e-5, atol=1e-9)


# ---- independent SEMANTIC pins for the CUDA-only rows (they do not pin the oracle's rounding; they break the loop of one
# author's transcription being checked against itself: each compares the HIP operator with a different formulation of
# what the reference kernel is FOR, computed in fp64 / integers with numpy, outside a mask of genuinely ambiguous inputs)

def test_a10_semantic_pin_fp64_nearest_neighbour(cuda):
    """A10: index == fp64 numpy argmin of the squared distance, for every query whose runner-up is not within 1e-6
    (relative) of the winner."""
    from deftet_amd import hip_ops
    rng = np.random.default_rng(10)
    q = (rng.random((2, 3000, 3)) - 0.5).astype(np.float32)
    d = rng.standard_normal((2, 20000, 3))
    pts = (0.35 * d / np.linalg.norm(d, axis=-1, keepdims=True) + 0.01 * rng.standard_normal((2, 20000, 3))).astype(np.float32)
    got = hip_ops.nn_index(torch.from_numpy(q).to(cuda), torch.from_numpy(pts).to(cuda)).cpu().numpy()
    for b in range(2):
        d2 = ((q[b].astype(np.float64)[:, None, :] - pts[b].astype(np.float64)[None, :, :]) ** 2).sum(-1)      # [3000, 20000]
        order = np.argpartition(d2, 1, axis=1)[:, :2]
        best = np.take_along_axis(d2, order, 1)
        first = np.where(best[:, 0] <= best[:, 1], order[:, 0], order[:, 1])
        lo, hi = best.min(1), best.max(1)
        clear = (hi - lo) > 1e-6 * hi
        assert clear.mean() > 0.99
        assert np.array_equal(got[b][clear], first[clear])


def _true_point_triangle_d2(p, tri):
    """fp64 squared distance from points p [P,3] to triangles tri [F,3,3] (Ericson's closest point), [P,F]."""
    a, b, c = tri[None, :, 0], tri[None, :, 1], tri[None, :, 2]
    p = p[:, None, :]
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
    bp = p - b
    d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
    cp = p - c
    d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    with np.errstate(divide="ignore", invalid="ignore"):
        denom = 1.0 / (va + vb + vc)
        v_in, w_in = vb * denom, vc * denom
        t_ab = d1 / (d1 - d3)
        t_ac = d2 / (d2 - d6)
        t_bc = (d4 - d3) / ((d4 - d3) + (d5 - d6))
    cl = a + ab * v_in[..., None] + ac * w_in[..., None]
    def put(mask, val):
        nonlocal cl
        cl = np.where(mask[..., None], val, cl)
    put((vb <= 0) & (d2 >= 0) & (d6 <= 0), a + ac * t_ac[..., None])
    put((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), b + (c - b) * t_bc[..., None])
    put((vc <= 0) & (d1 >= 0) & (d3 <= 0), a + ab * t_ab[..., None])
    put((d6 >= 0) & (d5 <= d6), np.broadcast_to(c, cl.shape))
    put((d3 >= 0) & (d4 <= d3), np.broadcast_to(b, cl.shape))
    put((d1 <= 0) & (d2 <= 0), np.broadcast_to(a, cl.shape))
    return ((p - cl) ** 2).sum(-1)


def test_a9_semantic_pin_true_point_triangle_distance(cuda, oracle):
    """A9: on a surface WITHOUT the faces for which the reference's xy-only inside test is degenerate (nearly vertical
    ones), closest_d == the true fp64 point-to-triangle squared distance (Ericson), and closest_f attains it."""
    from deftet_amd import hip_ops, surface_losses as SL
    v, faces = _sphere_surfaces(cuda, oracle, [0.33])
    tri = SL.corners(v, faces[0][None])[0].cpu().numpy().astype(np.float64)
    # a generic rotation, then drop the faces whose normal is within ~12 degrees of the xy plane
    rng = np.random.default_rng(9)
    qm, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    tri = tri @ qm.T
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    keep = np.abs(n[:, 2]) > 0.2 * np.linalg.norm(n, axis=1)
    tri = tri[keep]
    assert tri.shape[0] > 150
    p = (rng.random((2000, 3)) - 0.5) * 0.9
    t32 = torch.from_numpy(tri.astype(np.float32))[None].to(cuda)
    p32 = torch.from_numpy(p.astype(np.float32))[None].to(cuda)
    nf = torch.tensor([float(tri.shape[0])], device=cuda)
    d, f = hip_ops.tri_dist_fwd(p32, t32, nf)
    d, f = d[0, :, 0].cpu().numpy().astype(np.float64), f[0, :, 0].cpu().numpy().astype(np.int64)
    true = _true_point_triangle_d2(p32[0].cpu().numpy().astype(np.float64), t32[0].cpu().numpy().astype(np.float64))
    best = true.min(1)
    assert np.abs(d - best).max() <= 2e-4 * best.max() + 1e-9
    assert np.abs(true[np.arange(p.shape[0]), f] - best).max() <= 2e-4 * best.max() + 1e-9


def test_a8_semantic_pin_integer_edge_adjacency(cuda, oracle):
    """A8 (matching by POSITION) == adjacency by shared vertex-id pairs on an indexed surface whose vertices are distinct
    points: same neighbour sets, ascending, for every face."""
    from deftet_amd import hip_ops, surface_losses as SL
    v, faces = _sphere_surfaces(cuda, oracle, [0.3])
    f = faces[0].cpu().numpy()
    tri = SL.corners(v, faces[0][None])[0].contiguous()
    adj = hip_ops.face_edge_adj(tri, 30).cpu().numpy().astype(np.int64)
    edges = {}
    for i, (a, b, c) in enumerate(f):
        for e in ((a, b), (b, c), (c, a)):
            edges.setdefault((min(e), max(e)), []).append(i)
    want = [set() for _ in range(f.shape[0])]
    for fs in edges.values():
        for i in fs:
            want[i].update(j for j in fs if j != i)
    for i in range(f.shape[0]):
        row = adj[i][adj[i] >= 0].tolist()
        assert row == sorted(want[i]), i



def test_tri_dist_backward_with_forward_order_matches_atomic_and_sorted_paths(cuda, oracle):
    """The grouped backward (atomics per distinct face of a wavefront, points walked in the forward's grid order) gives
    the same face gradient as the per-point atomic path and the deterministic sorted path, up to fp32 summation order."""
    from deftet_amd import hip_ops
    radii = [0.3, 0.22, 0.38]
    v, faces = _sphere_surfaces(cuda, oracle, radii, res=24)
    Fmax = max(f.shape[0] for f in faces)
    face = torch.zeros(len(faces), Fmax, 3, 3, device=cuda)
    for b, f in enumerate(faces):
        face[b, : f.shape[0]] = v[b][f]
    nfb = torch.tensor([float(f.shape[0]) for f in faces], device=cuda)
    rng = np.random.default_rng(77)
    d3 = rng.standard_normal((len(radii), 20000, 3))
    pts = torch.from_numpy((d3 / np.linalg.norm(d3, axis=2, keepdims=True) * np.array(radii)[:, None, None]).astype(np.float32)).to(cuda)
    d, f, order = hip_ops.tri_dist_fwd(pts, face, nfb, want_order=True)
    assert order is not None and order.dtype == torch.int32 and order.shape == pts.shape[:2]
    srt = torch.sort(order.long(), dim=1).values                       # a permutation of the points of every shape
    assert torch.equal(srt, torch.arange(pts.shape[1], device=cuda).expand_as(srt))
    d2, f2 = hip_ops.tri_dist_fwd(pts, face, nfb)
    assert torch.equal(d, d2) and torch.equal(f, f2)
    g = torch.rand_like(d)
    a = hip_ops.tri_dist_bwd(pts, face, f, g)
    s = hip_ops.tri_dist_bwd(pts, face, f, g, deterministic=True)
    o = hip_ops.tri_dist_bwd(pts, face, f, g, order=order)
    scale = s.abs().max().item()
    assert scale > 0
    assert (o - s).abs().max().item() <= 2e-5 * scale and (a - s).abs().max().item() <= 2e-5 * scale
    # through the reference-shaped autograd function (which asks for the order when the faces need a gradient)
    from deftet_amd.layers.DefTet.tet_analytic_distance_batch.utils import tet_analytic_distance_f_batch
    fr = face.clone().requires_grad_(True)
    dd, _ = tet_analytic_distance_f_batch(pts, fr, nfb)
    (dd * g).sum().backward()
    assert (fr.grad - s).abs().max().item() <= 2e-5 * scale


def test_chamfer_to_cloud_matches_the_torch_composition(cuda, oracle):
    """hip_ops.chamfer_to_cloud (sample placement + distance + gradient as three HIP launches around A10) == the same
    quantity written with torch ops on the same random numbers: sample_on_faces -> nn_index -> gather -> sqrt -> masked
    sum, value and gradient w.r.t. the corners (fp64 autograd of the torch expression as the gradient reference)."""
    from deftet_amd import hip_ops
    from deftet_amd import surface_losses as SL
    radii = [0.3, 0.22, 0.0, 0.38]                                    # one empty surface
    v, faces = _sphere_surfaces(cuda, oracle, radii, res=16)
    counts = [int(f.shape[0]) for f in faces]
    Fmax, K, B = max(counts), 7, len(faces)
    idxs = torch.nn.utils.rnn.pad_sequence([f.long() for f in faces], batch_first=True)
    tri = SL.corners(v, idxs).clone().requires_grad_(True)
    rng = np.random.default_rng(5)
    gt = torch.from_numpy((rng.standard_normal((B, 3000, 3)) * 0.2).astype(np.float32)).to(cuda)
    gen = torch.Generator(device=cuda)
    gen.manual_seed(1234)
    out = hip_ops.chamfer_to_cloud(tri, gt, counts, K, gen)
    w = torch.rand(B, device=cuda) + 0.5
    (out * w).sum().backward()
    # the same with torch ops (same generator state -> same r)
    gen.manual_seed(1234)
    tri2 = tri.detach().clone().requires_grad_(True)
    samples = SL.sample_on_faces(tri2, K, gen).reshape(B, -1, 3)
    idx = hip_ops.nn_index_ragged(samples.detach(), gt, [c * K for c in counts]).long()
    near = torch.gather(gt, 1, idx[..., None].expand(-1, -1, 3))
    ok = (torch.arange(Fmax * K, device=cuda)[None, :] < (torch.tensor(counts, device=cuda) * K)[:, None])
    d = torch.sqrt(((samples - near) ** 2).sum(-1) + 1e-10)
    ref = (d * ok).sum(-1)
    (ref * w).sum().backward()
    assert torch.allclose(out, ref, rtol=2e-6, atol=1e-6)
    assert out[2].item() == 0.0                                       # the empty surface contributes nothing
    scale = tri2.grad.abs().max().item()
    assert scale > 0 and (tri.grad - tri2.grad).abs().max().item() <= 2e-5 * scale
    assert torch.equal(tri.grad[2], torch.zeros_like(tri.grad[2]))
