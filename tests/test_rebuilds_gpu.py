"""N3: render-side geometry rebuilds of
diff_render/diftet_6_subdiv/3_model/prepare_for_wz.py — integer work, bit-exact against the
reference's own outputs (tests/golden/n3_rebuilds.npz) and against the oracle at larger sizes."""
import os

import numpy as np
import pytest
import torch

from deftet_amd import grids

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "n3_rebuilds.npz")


def _gold(name):
    g = np.load(GOLD)
    return {k[len(name) + 1:]: g[k] for k in g.files if k.startswith(name + "_")}


@pytest.mark.parametrize("name", ["grid", "soup"])
def test_rebuilds_match_reference_outputs(cuda, name):
    from deftet_amd.render import prepare_for_wz as W
    G = _gold(name)
    t, P = G["tet"], int(G["n_point"])
    e = W.generate_edge(t)
    assert np.array_equal(e, G["edges"])
    assert np.array_equal(W.generate_tet_edge_idx(t, e), G["tet_edge"])
    from deftet_amd import hip_ops
    td = torch.from_numpy(t).to(cuda)
    ed, ted = hip_ops.tet_edges(td, P)
    assert np.array_equal(ed.cpu().numpy(), G["edges"]) and np.array_equal(ted.cpu().numpy(), G["tet_edge"])
    pn, fn, tn = W.generate_subdivision(t, G["pts"], G["feat"])
    assert np.array_equal(pn, G["sub_pts"]) and np.array_equal(fn, G["sub_feat"]) and np.array_equal(tn, G["sub_tet"])
    pn2, fn2, tn2 = W.generate_subdivision(t, G["pts"], G["feat"], G["sig"])
    assert np.array_equal(pn2, G["sub_pts_sig"]) and np.array_equal(tn2, G["sub_tet_sig"])
    mp, mf = W.generate_edge_points(G["pts"], G["feat"], G["edges"])
    assert np.array_equal(mp, G["sub_pts"][P:]) and np.array_equal(mf, G["sub_feat"][P:])
    table, adjsum = W.generate_point_adj_idx(P, t)
    assert table.dtype == np.int64 and adjsum.dtype == np.float32
    assert np.array_equal(table, G["adj_table"]) and np.array_equal(adjsum, G["adjsum"])
    assert np.array_equal(W.delete_tet(t, G["weights"], 0.01), G["kept"])
    assert np.array_equal(W.tetweights2tetneighbourweights(G["weights"], G["nei"], 1), G["nw1"], equal_nan=True)
    assert np.array_equal(W.tetweights2tetneighbourweights(G["weights"], G["nei"], 2), G["nw2"], equal_nan=True)


@pytest.mark.parametrize("res", [10, 40])
def test_rebuilds_vs_oracle_larger(cuda, oracle, res):
    from deftet_amd import hip_ops
    verts, tets = grids.kuhn_grid(res)
    rng = np.random.default_rng(res)
    t = tets.astype(np.int64)[rng.permutation(len(tets))]
    P = verts.shape[0]
    pts = rng.standard_normal((P, 3)).astype(np.float32)
    feat = rng.standard_normal((P, 7)).astype(np.float32)
    sig = rng.random(len(t)) < 0.5
    td, pd, fd = torch.from_numpy(t).to(cuda), torch.from_numpy(pts).to(cuda), torch.from_numpy(feat).to(cuda)
    e, te = hip_ops.tet_edges(td, P)
    eo = oracle.generate_edge(t)
    assert np.array_equal(e.cpu().numpy(), eo) and np.array_equal(te.cpu().numpy(), oracle.generate_tet_edge_idx(t, eo))
    for s in (None, sig, np.zeros(len(t), bool), np.ones(len(t), bool)):
        got = hip_ops.subdivide(td, pd, fd, None if s is None else torch.from_numpy(s).to(cuda))
        want = oracle.generate_subdivision(t, pts, feat, s)
        for a, b in zip(got, want):
            assert np.array_equal(a.cpu().numpy(), b)
    # subdividing twice keeps the mesh consistent: every child edge is shared by the right number of tets
    p2, f2, t2 = hip_ops.subdivide(td, pd, fd)
    assert t2.shape[0] == 8 * len(t) and p2.shape[0] == P + eo.shape[0]
    assert int(t2.max().item()) == p2.shape[0] - 1
    table, adjsum = hip_ops.point_adj_idx(P, td)
    wt, ws = oracle.generate_point_adj_idx(P, t)
    assert np.array_equal(table.cpu().numpy(), wt) and np.array_equal(adjsum.cpu().numpy(), ws)
    w = (rng.random((len(t), 4)) * 0.02).astype(np.float32)
    assert np.array_equal(hip_ops.delete_tet(td, torch.from_numpy(w).to(cuda), 0.01).cpu().numpy(), oracle.delete_tet(t, w, 0.01))
    nei = rng.integers(-1, len(t), (len(t), 4)).astype(np.int64)
    got = hip_ops.tet_neighbour_weights(torch.from_numpy(w).to(cuda), torch.from_numpy(nei).to(cuda), 2)
    assert np.array_equal(got.cpu().numpy(), oracle.tetweights2tetneighbourweights(w, nei, 2))


def test_rebuild_edge_cases(cuda, oracle):
    from deftet_amd import hip_ops
    e0 = torch.zeros(0, 4, dtype=torch.int64, device=cuda)
    ed, te = hip_ops.tet_edges(e0, 5)
    assert ed.shape == (0, 2) and te.shape == (0, 6)
    pts = torch.randn(5, 3, device=cuda)
    pn, fn, tn = hip_ops.subdivide(e0, pts, pts)
    assert torch.equal(pn, pts) and tn.shape == (0, 4)
    table, adjsum = hip_ops.point_adj_idx(5, e0)
    assert table.shape == (5, 0) and (adjsum == 0).all()
    assert hip_ops.delete_tet(e0, torch.zeros(0, 4, device=cuda)).shape == (0, 4)
    with pytest.raises(IndexError):
        hip_ops.tet_edges(torch.tensor([[0, 1, 2, 9]], device=cuda), 5)
    # one tet: 6 edges, 8 children, every vertex has 3 neighbours
    one = torch.tensor([[3, 1, 0, 2]], device=cuda)
    ed, te = hip_ops.tet_edges(one, 4)
    assert ed.tolist() == [[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]]
    assert te.tolist() == [[4, 2, 5, 0, 3, 1]]
    table, adjsum = hip_ops.point_adj_idx(4, one)
    assert table.tolist() == [[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]] and adjsum.flatten().tolist() == [3.0] * 4
