"""N1: kal.ops.mesh.check_sign restated.  PARITY UNPINNED (Kaolin is not in the
reference tree): the HIP paths are bit-exact (crossing COUNTS, not only parity) against
oracle/deftet_oracle_sign.c and against each other, and pinned semantically: the centroid of a tet
is inside the closed boundary surface of a tet subset iff its tet belongs to the subset."""
import numpy as np
import pytest
import torch

from deftet_amd import grids

pytestmark = pytest.mark.gpu


def closed_surface(res, batch, r=0.3):
    """vertex positions [B,V,3], boundary faces [F,3] (indices) of the tets whose centroid of the
    UNJITTERED grid lies in a sphere, the tet list and the selection mask."""
    from oracle import oracle as O
    verts, tets = grids.kuhn_grid(res)
    pos = grids.jittered_positions(verts, res, batch).astype(np.float32)
    f3, t2, _, _, _ = O.tet_to_face(tets, verts.shape[0])
    occ = np.linalg.norm((verts - 0.5)[tets].mean(1), axis=1) < r
    sel = occ[t2].sum(1) == 1
    return pos, f3[sel].astype(np.int64), tets, occ


def _run(cuda, verts, faces, pts):
    from deftet_amd import hip_ops
    v, f, p = torch.from_numpy(verts).to(cuda), torch.from_numpy(faces).to(cuda), torch.from_numpy(pts).to(cuda)
    a, ca = hip_ops.check_sign(v, f, p, return_count=True)
    b, cb = hip_ops.check_sign(v, f, p, brute=True, return_count=True)
    return a.cpu().numpy(), ca.cpu().numpy(), b.cpu().numpy(), cb.cpu().numpy()


@pytest.mark.parametrize("res,batch", [(8, 1), (16, 3)])
def test_centroids_of_closed_surface(cuda, oracle, res, batch):
    pos, faces, tets, occ = closed_surface(res, batch)
    cen = pos[:, tets].mean(2).astype(np.float32)
    rng = np.random.default_rng(res)
    pts = np.concatenate([cen, (rng.random((batch, 3000, 3)) - 0.5).astype(np.float32)], axis=1)
    a, ca, b, cb = _run(cuda, pos, faces, pts)
    want, cw = oracle.check_sign(pos, faces, pts, return_count=True)
    assert np.array_equal(ca, cw) and np.array_equal(cb, cw)          # crossing counts, bit for bit
    assert np.array_equal(a, want) and np.array_equal(b, want)
    T = tets.shape[0]
    assert np.array_equal(a[:, :T], np.broadcast_to(occ, (batch, T)))  # semantic pin
    assert 0.05 < a[:, T:].mean() < 0.25                               # sphere r=0.3 in the unit cube: ~11 %


def _soup(seed):
    rng = np.random.default_rng(seed)
    V, F, N = 400, 900, 4000
    verts = (rng.random((2, V, 3)) - 0.5).astype(np.float32)
    faces = rng.integers(0, V, (F, 3)).astype(np.int64)
    faces[::11, 1] = faces[::11, 0]                                    # degenerate (repeated vertex)
    verts[:, 5] = np.nan
    verts[:, 6] = np.inf
    verts[:, 7] = 3.0e6                                                # huge
    verts[:, 8:40, 1] = 0.125                                          # faces in the plane y = const: edge-on to the ray
    faces[100:130] = rng.integers(8, 40, (30, 3))
    verts[1, 40:60] *= 1e-6                                            # tiny faces
    faces[200:215] = rng.integers(40, 60, (15, 3))
    faces[300] = [0, 1, 2]
    verts[:, 0] = [-0.5, -0.5, -0.5]; verts[:, 1] = [-0.5, 0.5, -0.4]; verts[:, 2] = [-0.5, -0.4, 0.5]   # one face covering everything
    pts = (1.2 * (rng.random((2, N, 3)) - 0.5)).astype(np.float32)
    pts[:, 0] = np.nan
    pts[:, 1] = [np.inf, 0.1, 0.1]
    pts[:, 2] = [0.1, -np.inf, 0.2]
    pts[:, 3] = [2.0e6, 0.0, 0.0]
    pts[:, 4] = [-2.0e6, 0.01, 0.02]
    pts[:, 10:200, 1:] = verts[:, faces[10:200, 0], 1:]                # rays through mesh vertices
    pts[:, 200:400] = 0.5 * (verts[:, faces[10:210, 0]] + verts[:, faces[10:210, 1]]) - np.float32([0.3, 0, 0])   # through edges
    pts[:, 400:500, 1] = 0.125                                         # in the plane of the edge-on faces
    return verts, faces, pts


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_adversarial_soup(cuda, oracle, seed):
    verts, faces, pts = _soup(seed)
    a, ca, b, cb = _run(cuda, verts, faces, pts)
    with np.errstate(all="ignore"):
        want, cw = oracle.check_sign(verts, faces, pts, return_count=True)
    assert np.array_equal(cb, cw)
    assert np.array_equal(ca, cw)
    assert np.array_equal(a, want) and np.array_equal(b, want)
    assert cw.max() >= 2


def test_edge_cases_and_module(cuda, oracle):
    from deftet_amd import hip_ops
    from deftet_amd.layers.DefTet.deftet import DefTet
    pos, faces, tets, occ = closed_surface(8, 2)
    v, f = torch.from_numpy(pos).to(cuda), torch.from_numpy(faces).to(cuda)
    p = torch.from_numpy(pos[:, tets].mean(2).astype(np.float32)).to(cuda)
    assert hip_ops.check_sign(v, f[:0], p).sum() == 0                  # no faces: all outside
    assert hip_ops.check_sign(v, f, p[:, :0]).shape == (2, 0)
    with pytest.raises(IndexError):
        hip_ops.check_sign(v, torch.tensor([[0, 1, 10 ** 6]], device=cuda), p)
    # DefTet.check_tet_inside_sdfs (layers/DefTet/deftet.py:33-49): mesh_list = (verts list, [faces] list)
    m = DefTet(device=cuda)
    tet = torch.from_numpy(pos[:, tets]).to(cuda)
    occ_m = m.check_tet_inside_sdfs(tet, ([v[0:1], v[1:2]], [[f], [f]]))
    assert occ_m.shape == (2, tets.shape[0], 1) and occ_m.dtype == torch.float32
    assert np.array_equal(occ_m[..., 0].cpu().numpy() > 0.5, np.broadcast_to(occ, (2, tets.shape[0])))


def test_grid_equals_brute_full_size(cuda):
    """BASELINE-sized call: all 257,250 centroids of the res-70 grid against the sphere surface."""
    from deftet_amd import hip_ops
    pos, faces, tets, occ = closed_surface(70, 2)
    v, f = torch.from_numpy(pos).to(cuda), torch.from_numpy(faces).to(cuda)
    p = torch.from_numpy(pos[:, tets].mean(2).astype(np.float32)).to(cuda)
    a, ca = hip_ops.check_sign(v, f, p, return_count=True)
    b, cb = hip_ops.check_sign(v, f, p, brute=True, return_count=True)
    assert torch.equal(ca, cb)
    assert np.array_equal(a.cpu().numpy(), np.broadcast_to(occ, (2, tets.shape[0])))


def test_ragged_batch_of_different_meshes(cuda, oracle):
    """a different ground-truth mesh per shape (layers/DefTet/deftet.py:44-47) in one launch sequence:
    crossing counts equal the per-shape oracle, for the binned and the brute path"""
    from deftet_amd import hip_ops
    from deftet_amd.layers.DefTet.deftet import DefTet
    meshes = []
    for res, r in ((8, 0.3), (12, 0.25), (6, 0.35)):
        pos, faces, tets, occ = closed_surface(res, 1, r)
        meshes.append((pos[0], faces, tets, occ))
    sv, sf, _ = _soup(3)
    meshes.append((sv[0], sf, None, None))
    B, N = len(meshes), 3000
    rng = np.random.default_rng(9)
    pts = (rng.random((B, N, 3)) - 0.5).astype(np.float32)
    pts[3, :5] = np.nan
    vl = [torch.from_numpy(m[0]).to(cuda) for m in meshes]
    fl = [torch.from_numpy(m[1]).to(cuda) for m in meshes]
    p = torch.from_numpy(pts).to(cuda)
    for brute in (False, True):
        got, cnt = hip_ops.check_sign_ragged(vl, fl, p, brute=brute, return_count=True)
        for b, m in enumerate(meshes):
            with np.errstate(all="ignore"):
                want, cw = oracle.check_sign(m[0][None], m[1], pts[b:b + 1], return_count=True)
            assert np.array_equal(cnt[b].cpu().numpy(), cw[0]), (brute, b)
            assert np.array_equal(got[b].cpu().numpy(), want[0])
    with pytest.raises(IndexError):
        hip_ops.check_sign_ragged(vl, [fl[0], fl[1], fl[2], torch.tensor([[0, 1, 10 ** 6]], device=cuda)], p, check=True)
    # the DefTet mirror with per-shape meshes: centroids of each shape's own grid
    m3 = meshes[:3]
    T = min(x[2].shape[0] for x in m3)
    tet = torch.stack([torch.from_numpy(x[0][x[2][:T]]) for x in m3]).to(cuda)
    occ_m = DefTet(device=cuda).check_tet_inside_sdfs(tet, ([v[None] for v in vl[:3]], [[f] for f in fl[:3]]))
    assert occ_m.shape == (3, T, 1)
    for b, x in enumerate(m3):
        assert np.array_equal(occ_m[b, :, 0].cpu().numpy() > 0.5, x[3][:T])
