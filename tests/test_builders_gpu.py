"""GPU parity tests for the adjacency / face-table builders (A2-A6): bit-exact against the
oracle, the golden vectors from the reference's Python twins and (through the reference-shaped
interface classes) the same canonical forms the reference consumes."""
import os

import numpy as np
import pytest
import torch

from deftet_amd import grids
from tests.test_cpu_oracle_golden import GOLD, SMALL, _random_mesh, lexsorted, load, split_share

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", SMALL + ["cube40"])
def test_builders_device_vs_oracle_and_golden(cuda, oracle, name):
    from deftet_amd import hip_ops
    if name == "cube40":
        g = load("cube40_grid.npz")
        tets, n_point = g["tets"], g["verts"].shape[0]
        gold = None
    else:
        gold = load("builders_%s.npz" % name)
        tets, n_point = gold["tets"], gold["verts"].shape[0]
    rows = hip_ops.tet_adj_share(tets, n_point, cuda).cpu().numpy()
    assert np.array_equal(rows, oracle.tet_adj_share(tets, n_point))                  # row order included
    for wrap in (True, False):
        fa = hip_ops.tet_face_adj(tets, n_point, cuda, wrap32=wrap).cpu().numpy()
        assert np.array_equal(fa, oracle.tet_face_adj(tets, n_point, wrap32=wrap))
    pa = hip_ops.tet_point_adj(tets, n_point, cuda).cpu().numpy()
    assert np.array_equal(pa, oracle.tet_point_adj(tets, n_point))
    for wb in (False, True):
        got = hip_ops.tet_to_face(tets, n_point, cuda, with_boundary=wb)
        want = oracle.tet_to_face(tets, n_point, with_boundary=wb)
        for a, b in zip(got[:4], want[:4]):
            assert np.array_equal(a.cpu().numpy(), b)
        assert got[4] == want[4]
    if gold is not None:
        assert np.array_equal(pa.astype(np.int64), lexsorted(gold["point_adj_idx"]))
        assert np.array_equal(lexsorted(fa).astype(np.int64), gold["face_adj_rows"])
        f3, t2, tf2, b3, _ = hip_ops.tet_to_face(tets, n_point, cuda)
        assert np.array_equal(f3.cpu().numpy(), gold["face_fx3"].reshape(-1, 3))
        assert np.array_equal(b3.cpu().numpy(), gold["boundary_fx3"])
        if not gold["adj_share_raises"][0]:
            for i in range(4):
                assert np.array_equal(split_share(rows, i), gold["adj_share_%d" % i])


@pytest.mark.parametrize("seed", range(4))
def test_builders_random_meshes(cuda, oracle, seed):
    from deftet_amd import hip_ops
    rng = np.random.default_rng(100 + seed)
    tets, n_point = _random_mesh(rng, 4 + 2 * seed)
    if seed == 3:                                   # sparse vertex ids beyond 46340: int32 edge-key wrap
        remap = np.sort(rng.choice(90000, n_point, replace=False)).astype(np.int32)
        tets, n_point = remap[tets], 90000
    assert np.array_equal(hip_ops.tet_adj_share(tets, n_point, cuda).cpu().numpy(), oracle.tet_adj_share(tets, n_point))
    assert np.array_equal(hip_ops.tet_face_adj(tets, n_point, cuda, True).cpu().numpy(),
                          oracle.tet_face_adj(tets, n_point, True))
    assert np.array_equal(hip_ops.tet_point_adj(tets, n_point, cuda).cpu().numpy(), oracle.tet_point_adj(tets, n_point))
    got, want = hip_ops.tet_to_face(tets, n_point, cuda), oracle.tet_to_face(tets, n_point)
    for a, b in zip(got[:4], want[:4]):
        assert np.array_equal(a.cpu().numpy(), b)


def test_colaps_v_decimal_rounding(cuda, oracle):
    from deftet_amd import hip_ops
    rng = np.random.default_rng(5)
    pts = rng.uniform(-1, 1, (4000, 3)).astype(np.float32)
    pts[1000:2000] = pts[:1000] + rng.choice([0, 1e-7, 4e-6, 1e-5], (1000, 3)).astype(np.float32)
    pts[2000:2400] = np.round(pts[2000:2400], 5) + np.float32(5e-6)           # %.5f rounding ties
    k = np.arange(400, dtype=np.float64)
    pts[2400:2800, 0] = ((2 * k + 1) / 2 ** 7 * 1e-3).astype(np.float32)      # exactly representable ...5 ties
    pts[2800] = 0.0
    pts[2801] = -0.0
    pts[2802] = -1e-7                                                        # "-0.00000" != "0.00000"
    pts[2803] = [1e-7, -1e-7, 0]
    pts[2804] = [np.inf, -np.inf, np.nan]
    pts[2805] = [np.inf, -np.inf, np.nan]
    pts[2806] = [3e7, -3e7, 2 ** 24]
    pts[2807] = [3e7, -3e7, 2 ** 24]
    pts[2808] = [1e-45, -1e-45, 1e-38]
    pts[2809:2900] = (rng.uniform(-1, 1, (91, 3)) * 1e5).astype(np.float32)
    m, inv = hip_ops.colaps_v(torch.from_numpy(pts).to(cuda))
    wm, winv = oracle.colaps_v(pts)
    assert np.array_equal(m.cpu().numpy(), wm) and np.array_equal(inv.cpu().numpy(), winv)
    if oracle.RefBuilders.available():
        rm, rinv = oracle.RefBuilders().colaps_v(pts)
        assert np.array_equal(m.cpu().numpy(), rm) and np.array_equal(inv.cpu().numpy(), rinv)


def test_reference_shaped_interfaces(cuda, oracle):
    """the classes at the reference's import paths (host-pointer C ABI underneath)"""
    from deftet_amd.utils import tet_utils as tu
    from deftet_amd.utils.lib.colaps_v.interface import Tet_point_adj as Colaps
    g = load("builders_kuhn4.npz")
    tets, verts = g["tets"], g["verts"]
    n_point = verts.shape[0]
    adjn = tu.c_tet_to_adj_sparse(verts, tets, normalize=True).coalesce()
    assert np.array_equal(adjn.indices().numpy().T, g["point_adj_norm_idx"])
    assert np.allclose(adjn.values().numpy(), g["point_adj_norm_val"], rtol=1e-6)
    fa = tu.c_tet_to_face_adj_sparse(verts, tets).tocoo()
    o = np.lexsort((fa.col, fa.row))
    assert np.array_equal(np.stack([fa.row[o], fa.col[o]], 1), g["face_adj_rows"]) and (fa.data == 1).all()
    share = tu.c_tet_adj_share(tets, n_point)
    for i in range(4):
        assert np.array_equal(share[i].coalesce().indices().numpy().T, g["adj_share_%d" % i])
    f3, t2, tf2, b3 = tu.tet_to_face(n_point, tets)
    assert np.array_equal(f3, g["face_fx3"]) and np.array_equal(t2, g["face_tetidx_fx2"])
    assert np.array_equal(tf2, g["face_tetfaceidx_fx2"]) and np.array_equal(b3, g["boundary_fx3"])
    w3, w2, wf2 = tu.tet_to_face_idx(n_point, tets, with_boundary=True)
    assert np.array_equal(w3, g["facewb_fx3"]) and np.array_equal(w2, g["facewb_tetidx_fx2"])
    pts = (verts - 0.5).astype(np.float32)
    pts2 = np.concatenate([pts, pts[::3]], 0)
    m, inv = Colaps().run(pts2)
    wm, winv = oracle.colaps_v(pts2)
    assert np.array_equal(m, wm) and np.array_equal(inv, winv)


def test_builders_res70_properties(cuda):
    """BASELINE size (res=70, T=257,250): size-independent invariants of the outputs."""
    from deftet_amd import hip_ops
    verts, tets = grids.kuhn_grid(70)
    n_point, T = verts.shape[0], tets.shape[0]
    f3, t2, tf2, b3, nm = hip_ops.tet_to_face(tets, n_point, cuda)
    assert nm == 0 and 2 * f3.shape[0] + b3.shape[0] == 4 * T            # every tet-face is interior(x2) or boundary
    assert b3.shape[0] == 6 * 2 * 35 * 35                                # 2 triangles per boundary square
    rows = hip_ops.tet_adj_share(tets, n_point, cuda)
    assert rows.shape[0] == 2 * f3.shape[0]
    r = rows.long()
    assert (r[0::2, 0] == r[1::2, 1]).all() and (r[0::2, 1] == r[1::2, 0]).all() and (r[0::2, 0] < r[0::2, 1]).all()
    tt = torch.from_numpy(tets).to(cuda).long()
    assert torch.equal(t2[:, 0], torch.sort(t2[:, 0]).values)             # first-seen order is tet order
    # the face listed for (first tet, local face) really is that tet's local face
    idx = torch.tensor([[0, 1, 2], [1, 0, 3], [2, 3, 0], [3, 2, 1]], device=cuda)
    assert torch.equal(f3, torch.gather(tt[t2[:, 0]], 1, idx[tf2[:, 0]]))
    pa = hip_ops.tet_point_adj(tets, n_point, cuda).long()
    key = pa[:, 0] * n_point + pa[:, 1]
    assert (key[1:] > key[:-1]).all()                                     # sorted, unique
    rev = pa[:, 1] * n_point + pa[:, 0]
    assert torch.equal(torch.sort(rev).values, key)                       # symmetric
    fa = hip_ops.tet_face_adj(tets, n_point, cuda, wrap32=False)
    fb = hip_ops.tet_face_adj(tets, n_point, cuda, wrap32=True)
    assert fa.shape == fb.shape                                           # 46,656 points: wraps but does not collide
    ka = torch.sort(fa[:, 0].long() * (4 * T) + fa[:, 1].long()).values
    kb = torch.sort(fb[:, 0].long() * (4 * T) + fb[:, 1].long()).values
    assert torch.equal(ka, kb)


@pytest.mark.parametrize("name", ["two", "kuhn2", "kuhn4", "kuhn4perm", "kuhn8", "cube40"])
def test_neighbour_table_and_face_owner_table_vs_reference_outputs(cuda, name):
    """T x 4 `tet_neighbour_idx` (utils_tetsv.tet_adj_share, diff_render/.../utils_tetsv.py:16-75) and the 4T x 2
    owner table of tet_to_face_withtet (utils/tet_utils.py:259-300): bit-exact against what the reference
    functions returned (fixtures written by gen_golden.py; sha256 for the shipped cube_40 grid)."""
    import hashlib
    from deftet_amd import hip_ops
    from deftet_amd.utils import tet_utils as TU
    if name == "cube40":
        g = load("cube40_grid.npz")
        tets, n_point = g["tets"], g["verts"].shape[0]
        h = load("cube40_hashes.npz")
        nbr, own = hip_ops.tet_neighbours(tets, n_point, cuda, want_face_owners=True)
        for key, arr in (("adj_share_nbr_tx4", nbr), ("face_withtet_4tx2", own)):
            got = hashlib.sha256(np.ascontiguousarray(arr.cpu().numpy().astype(np.int64)).tobytes()).digest()
            assert got == h[key].tobytes(), key
        return
    gold = load("builders_%s.npz" % name)
    tets, n_point = gold["tets"], gold["verts"].shape[0]
    nbr, own = hip_ops.tet_neighbours(tets, n_point, cuda, want_face_owners=True)
    assert nbr.dtype == torch.int64 and np.array_equal(nbr.cpu().numpy(), gold["adj_share_nbr_tx4"])
    assert np.array_equal(own.cpu().numpy(), gold["face_withtet_4tx2"])
    assert np.array_equal(TU.tet_neighbour_table(tets, n_point), gold["adj_share_nbr_tx4"])
    assert np.array_equal(TU.tet_to_face_withtet(gold["verts"], tets), gold["face_withtet_4tx2"])
    # the table feeds tetweights2tetneighbourweights (N3) unchanged
    w = torch.rand(tets.shape[0], 3, device=cuda)
    out = hip_ops.tet_neighbour_weights(w, nbr, 1)
    assert out.shape[0] == tets.shape[0]


def test_neighbour_table_rejects_non_manifold(cuda):
    from deftet_amd import hip_ops
    tets = np.array([[0, 1, 2, 3], [0, 1, 2, 4], [0, 1, 2, 5]], np.int32)        # face (0,1,2) has three owners
    with pytest.raises(ValueError):
        hip_ops.tet_neighbours(tets, 6, cuda)
