"""CPU tests (no GPU): the oracle against the golden vectors generated from the reference's
Python code (tests/golden/gen_golden.py) and against the reference's native builders
compiled into oracle/_ref."""
import hashlib
import os

import numpy as np
import pytest

from deftet_amd import grids

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL = ["one", "two", "kuhn2", "kuhn4", "kuhn4perm", "kuhn8"]


def load(name):
    return np.load(os.path.join(GOLD, name))


def lexsorted(rows):
    rows = np.asarray(rows)
    if rows.size == 0:
        return rows.reshape(0, rows.shape[-1] if rows.ndim > 1 else 2)
    return rows[np.lexsort(tuple(rows[:, k] for k in range(rows.shape[1] - 1, -1, -1)))]


def split_share(rows, i):
    sel = rows[rows[:, 2] == i][:, :2].astype(np.int64)
    return lexsorted(sel)


@pytest.mark.parametrize("name", SMALL)
def test_builders_match_python_twins(oracle, name):
    g = load("builders_%s.npz" % name)
    tets, n_point = g["tets"], g["verts"].shape[0]
    # A4 vertex adjacency (set semantics; reference order is hash order)
    assert np.array_equal(oracle.tet_point_adj(tets, n_point).astype(np.int64), lexsorted(g["point_adj_idx"]))
    # A3 face adjacency, compared as the canonical CSR the reference consumes
    for wrap in (True, False):
        assert np.array_equal(lexsorted(oracle.tet_face_adj(tets, n_point, wrap32=wrap)).astype(np.int64), g["face_adj_rows"])
    assert (g["face_adj_vals"] == 1).all()
    # A2 tet adjacency through shared faces
    rows = oracle.tet_adj_share(tets, n_point)
    if g["adj_share_raises"][0]:
        assert rows.shape[0] == 0
    else:
        for i in range(4):
            assert np.array_equal(split_share(rows, i), g["adj_share_%d" % i])
    # A6 face tables: bit-exact including row order and winding
    f3, t2, tf2, b3, nm = oracle.tet_to_face(tets, n_point, with_boundary=False)
    assert nm == 0
    assert np.array_equal(f3, g["face_fx3"].reshape(-1, 3)) and np.array_equal(t2, g["face_tetidx_fx2"].reshape(-1, 2))
    assert np.array_equal(tf2, g["face_tetfaceidx_fx2"].reshape(-1, 2)) and np.array_equal(b3, g["boundary_fx3"])
    f3, t2, tf2, _, _ = oracle.tet_to_face(tets, n_point, with_boundary=True)
    assert np.array_equal(f3, g["facewb_fx3"]) and np.array_equal(t2, g["facewb_tetidx_fx2"])
    assert np.array_equal(tf2, g["facewb_tetfaceidx_fx2"])


def test_builders_cube40_hashes(oracle):
    """the shipped QuarTet grid (data fixture cube40_grid.npz) against sha256 of the reference outputs"""
    g = load("cube40_grid.npz")
    h = load("cube40_hashes.npz")
    tets, n_point = g["tets"], g["verts"].shape[0]

    def sha(a):
        return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)

    pa = oracle.tet_point_adj(tets, n_point).astype(np.int64)
    assert pa.shape[0] == 118566 and np.array_equal(sha(pa), h["point_adj_idx"])
    f3, t2, tf2, b3, _ = oracle.tet_to_face(tets, n_point)
    assert np.array_equal(sha(f3), h["face_fx3"]) and np.array_equal(sha(t2), h["face_tetidx_fx2"])
    assert np.array_equal(sha(tf2), h["face_tetfaceidx_fx2"]) and np.array_equal(sha(b3), h["boundary_fx3"])
    assert f3.shape[0] == 92604 and b3.shape[0] == 4680
    rows = oracle.tet_adj_share(tets, n_point)
    for i in range(4):
        assert np.array_equal(sha(split_share(rows, i)), h["adj_share_%d" % i])
    fa = lexsorted(oracle.tet_face_adj(tets, n_point, wrap32=True)).astype(np.int64)
    assert fa.shape[0] == 4543344 and np.array_equal(sha(fa), h["face_adj_rows"])
    f3, t2, tf2, _, _ = oracle.tet_to_face(tets, n_point, with_boundary=True)
    assert np.array_equal(sha(f3), h["facewb_fx3"]) and np.array_equal(sha(t2), h["facewb_tetidx_fx2"])


def _random_mesh(rng, res):
    verts, tets = grids.kuhn_grid(res)
    tets = tets[rng.permutation(tets.shape[0])]
    relabel = rng.permutation(verts.shape[0]).astype(np.int32)
    tets = relabel[tets]
    for t in tets:                      # random local re-orderings (orientation is irrelevant to the builders)
        rng.shuffle(t)
    keep = rng.random(tets.shape[0]) > 0.2
    return tets[keep].astype(np.int32), verts.shape[0]


@pytest.mark.parametrize("seed", range(6))
def test_builders_match_reference_native(oracle, seed):
    """row-for-row equality with the reference's run.cpp compiled into oracle/_ref"""
    if not oracle.RefBuilders.available():
        pytest.skip("oracle/_ref not built (reference tree absent)")
    ref = oracle.RefBuilders()
    rng = np.random.default_rng(seed)
    tets, n_point = _random_mesh(rng, 4 + 2 * (seed % 3))
    assert np.array_equal(oracle.tet_adj_share(tets, n_point), ref.tet_adj_share(tets, n_point))
    assert np.array_equal(oracle.tet_face_adj(tets, n_point, wrap32=True), ref.tet_face_adj(tets, n_point))
    assert np.array_equal(oracle.tet_point_adj(tets, n_point), lexsorted(ref.tet_point_adj(tets, n_point)))
    pts = rng.uniform(-1, 1, (500, 3)).astype(np.float32)
    pts[100:200] = pts[:100] + rng.choice([0, 1e-7, 4e-6, 1e-5], (100, 3)).astype(np.float32)
    pts[200:220] = np.round(pts[200:220], 5) + np.float32(5e-6)        # rounding ties of %.5f
    pts[220] = 0.0
    pts[221] = -0.0                                                     # "-0.00000" != "0.00000"
    pts[222] = -1e-7
    a, b = oracle.colaps_v(pts), ref.colaps_v(pts)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_face_adj_int32_wrap_matches_native(oracle):
    """the native tet_face_adj keys edges with a 32-bit int (run.cpp:39): with n_point > 46340
    the key wraps; the oracle's wrap32 mode must still equal the native output row for row"""
    if not oracle.RefBuilders.available():
        pytest.skip("oracle/_ref not built")
    ref = oracle.RefBuilders()
    rng = np.random.default_rng(3)
    tets, n_small = _random_mesh(rng, 4)
    n_point = 70000
    remap = np.sort(rng.choice(n_point, n_small, replace=False)).astype(np.int32)
    tets = remap[tets]
    assert np.array_equal(oracle.tet_face_adj(tets, n_point, wrap32=True), ref.tet_face_adj(tets, n_point))
    # and the Python-twin mode (exact keys) gives the same adjacency as on the un-remapped mesh
    a = lexsorted(oracle.tet_face_adj(tets, n_point, wrap32=False))
    b = lexsorted(oracle.tet_face_adj(tets, n_point, wrap32=True))
    assert a.shape == b.shape          # no collision below 2^32


@pytest.mark.parametrize("name", ["kuhn4", "kuhn8"])
def test_bary_matches_reference_python(oracle, name):
    g = load("bary_%s.npz" % name)
    n = g["pts"].shape[0]
    tet = g["tet"][None]                       # treat each (tet_i, p_i) pair as tet i of one shape
    cond = np.arange(n, dtype=np.float32)[None]
    w = oracle.bary(tet, g["pts"][None], cond)[0]
    assert np.abs(w - g["w_f32"]).max() <= 1e-5 * np.abs(g["w_f64"]).max()
    w64, gt64 = oracle.point_in_tet_bwd_torch(tet, g["pts"][None], cond, g["grad_w"][None])
    assert np.allclose(w64[0], g["w_f64"], rtol=1e-12, atol=1e-12)
    assert np.allclose(gt64[0], g["grad_tet_f64"], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("res,jit", [(4, 0.1), (8, 0.15)])
def test_point_in_tet_semantics_vs_reference_barycentrics(oracle, res, jit):
    """Semantic pin of the (otherwise unpinned) point-in-tet restatement: on a valid mesh the
    tet it returns must contain the query according to the reference's own barycentric
    formula, and a query it rejects must not be strictly inside any tet."""
    tet, pts, _, _ = grids.make_case(res, 3000, 2, jit)
    cond = oracle.point_in_tet(tet, pts)
    tol = 1e-5
    for b in range(tet.shape[0]):
        import torch
        t64 = torch.from_numpy(tet[b]).double()
        p64 = torch.from_numpy(pts[b]).double()
        W = torch.stack(oracle.bary_torch(t64[None, :, 0], t64[None, :, 1], t64[None, :, 2], t64[None, :, 3],
                                          p64[:, None, :]), -1).numpy()          # [Q,T,4]
        inside = (W > tol).all(-1)                                             # strictly inside
        near = (W > -tol).all(-1)                                              # inside or within tol of the boundary
        c = cond[b, :, 0].astype(int)
        hit = c >= 0
        assert near[np.arange(len(c))[hit], c[hit]].all()
        assert not inside[~hit].any()
        first_inside = np.where(inside.any(1), inside.argmax(1), -1)
        strict = inside.any(1)
        # where some tet strictly contains the query, the answer is that tet unless a
        # lower-index tet touches the query within tol
        lower_ok = (c[strict] == first_inside[strict]) | (c[strict] < first_inside[strict])
        assert lower_ok.all()
    m = oracle.point_in_tet_margin(tet[0], pts[0][:200])
    assert (m >= 0).all()


@pytest.mark.parametrize("name", ["two", "kuhn2", "kuhn4", "kuhn4perm", "kuhn8"])
def test_neighbour_tables_oracle_matches_reference_outputs(oracle, name):
    """T x 4 tet_neighbour_idx (utils_tetsv.py:16-75) and 4T x 2 tet_to_face_withtet (utils/tet_utils.py:259-300)
    restated from the unique-face table == the reference functions' own outputs (gen_golden.py)."""
    gold = load("builders_%s.npz" % name)
    nbr, owners = oracle.tet_neighbours(gold["tets"], gold["verts"].shape[0])
    assert np.array_equal(nbr, gold["adj_share_nbr_tx4"])
    assert np.array_equal(owners, gold["face_withtet_4tx2"])


def test_surface_glue_normals_match_reference_mesh_utils():
    """deftet_amd/surface_losses.py corners / unit_normals (pure torch) == the reference's get_normal
    (utils/mesh_utils.py:42-52) on the boundary faces of a jittered grid (tests/golden/surface_glue.npz)."""
    import torch
    from deftet_amd import surface_losses as SL
    g = load("surface_glue.npz")
    v = torch.from_numpy(g["verts"])
    f = torch.from_numpy(g["faces"])[None]
    tri = SL.corners(v, f)
    assert torch.equal(tri[0], v[0][f[0]])
    n = SL.unit_normals(tri)[0].numpy()
    assert np.abs(n - g["normals"]).max() <= 1e-6
    # the loss value itself from the stored adjacency pairs (what normal_consistency computes once the A8 operator has produced them)
    pairs = torch.from_numpy(g["pairs"])
    nn_ = SL.unit_normals(tri)
    loss = (1.0 - (nn_[:, pairs[0]] * nn_[:, pairs[1]]).sum(-1)).mean(-1).numpy()
    assert np.abs(loss - g["normal_loss"]).max() <= 1e-6


def _pit_fixture(name):
    g = load("pit_index_%s.npz" % name)
    if name == "cube40":                                   # tets rebuilt from the shipped grid (train_multigpu.py:65-66 shift)
        c = load("cube40_grid.npz")
        tet = (c["verts"] - 0.5).astype(np.float32)[c["tets"]]
    else:
        tet = g["tet"]
    return tet, g["pts"], g["expected"], g["ambiguous"], g["w_ref_f32"]


@pytest.mark.parametrize("name", ["kuhn4", "kuhn8", "kuhn20", "soup", "cube40"])
def test_point_in_tet_index_pinned_by_reference_barycentrics(oracle, name):
    """A1 pin: tests/golden/pit_index_*.npz hold, for every query, the lowest tet index whose four
    weights from the reference's own bary_centric_tet (utils/tet_utils.py:28-45, imported and run on
    ALL T x Q pairs by gen_golden.py) exceed 1e-4, and a mask of the queries that touch a tet at or
    below that index within 1e-4.  Outside that mask the restatement of
    check_condition_tet_for.cu:105-189 must return exactly that index (or -1)."""
    tet, pts, expected, ambiguous, w_ref = _pit_fixture(name)
    got = oracle.point_in_tet(tet[None], pts[None])[0, :, 0].astype(np.int64)
    clear = ~ambiguous
    assert clear.mean() > 0.99
    assert np.array_equal(got[clear], expected[clear].astype(np.int64))
    # ambiguous queries: whatever the fp32 tests decide, the answer is a tet that touches the query
    # (checked through the weights of that tet) or -1
    w = oracle.bary(tet[None], pts[None], got[None].astype(np.float32))[0]
    hit = got >= 0
    assert (w[hit].min(-1) > -2e-4).all()
    # the A1b weights of the pinned tet agree with the reference's fp32 evaluation
    sel = clear & (expected >= 0)
    assert np.abs(w[sel] - w_ref[sel]).max() <= 1e-5 * max(1.0, np.abs(w_ref[sel]).max())


def test_deftet_module_fixture_is_selfconsistent():
    """deftet_module.npz pins the reference's own outputs for A7 / paste_occ / A11; here only
    shapes and invariants are checked on CPU (the HIP counterparts are tested with -m gpu)."""
    g = load("deftet_module.npz")
    occ = g["occ"]
    for i in range(occ.shape[0]):
        o2 = occ[i][g["tetidx_fx2"]]
        assert g["boundary_%d" % i].shape[0] == int((o2.sum(-1) == 1).sum())
        assert g["internal_%d" % i].shape[0] == int((o2.sum(-1) == 2).sum())
    c = g["cond"].copy()
    c[c < 0] = 0
    assert np.array_equal(g["cond_after"], c)
    assert np.array_equal(g["pasted"], np.take_along_axis(g["pred"], c[..., 0].astype(int), 1))


def test_tet_gather_oracle_matches_torch_autograd():
    """N2: the numpy restatement of the vertex->tet gather and of its backward equals
    torch.gather / torch autograd (the expression at layers/DefTet/deftet.py:65-68)."""
    import torch
    from oracle import oracle
    rng = np.random.default_rng(3)
    B, V, T = 3, 40, 300
    pos = rng.standard_normal((B, V, 3)).astype(np.float32)
    idx = rng.integers(0, V, (B, T, 4))
    g = rng.standard_normal((B, T, 4, 3)).astype(np.float32)
    p = torch.tensor(pos, dtype=torch.float64, requires_grad=True)
    out = torch.gather(p.unsqueeze(2).expand(-1, -1, 4, -1), 1, torch.tensor(idx).unsqueeze(-1).expand(-1, -1, -1, 3))
    assert np.array_equal(out.detach().numpy().astype(np.float32), oracle.tet_gather(pos, idx))
    out.backward(torch.tensor(g, dtype=torch.float64))
    got = oracle.tet_gather_bwd(g, idx, V)
    assert np.abs(p.grad.numpy() - got).max() <= 1e-5 * np.abs(p.grad.numpy()).max()
    # shared topology broadcast
    assert np.array_equal(oracle.tet_gather(pos, idx[0]), oracle.tet_gather(pos, np.broadcast_to(idx[0], idx.shape)))


@pytest.mark.parametrize("name", ["grid", "soup"])
def test_n3_rebuild_oracle_matches_reference_outputs(name):
    """N3: the vectorised restatements equal what prepare_for_wz.py / 3_model/deftet.py produced
    (tests/golden/n3_rebuilds.npz, written by tests/golden/gen_golden.py from the reference)."""
    from oracle import oracle
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "n3_rebuilds.npz"))
    G = {k[len(name) + 1:]: g[k] for k in g.files if k.startswith(name + "_")}
    t, P = G["tet"], int(G["n_point"])
    e = oracle.generate_edge(t)
    assert np.array_equal(e, G["edges"])
    assert np.array_equal(oracle.generate_tet_edge_idx(t, e), G["tet_edge"])
    pn, fn, tn = oracle.generate_subdivision(t, G["pts"], G["feat"])
    assert np.array_equal(pn, G["sub_pts"]) and np.array_equal(fn, G["sub_feat"]) and np.array_equal(tn, G["sub_tet"])
    pn2, _, tn2 = oracle.generate_subdivision(t, G["pts"], G["feat"], G["sig"])
    assert np.array_equal(pn2, G["sub_pts_sig"]) and np.array_equal(tn2, G["sub_tet_sig"])
    table, adjsum = oracle.generate_point_adj_idx(P, t)
    assert np.array_equal(table, G["adj_table"]) and np.array_equal(adjsum, G["adjsum"]) and adjsum.dtype == G["adjsum"].dtype
    with np.errstate(invalid="ignore"):
        assert np.array_equal(oracle.delete_tet(t, G["weights"], 0.01), G["kept"])
    assert np.array_equal(oracle.tetweights2tetneighbourweights(G["weights"], G["nei"], 1), G["nw1"], equal_nan=True)
    assert np.array_equal(oracle.tetweights2tetneighbourweights(G["weights"], G["nei"], 2), G["nw2"], equal_nan=True)


# ---------------------------------------------------------------------------- render-side glue vs the reference's own functions
def _render_glue():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "render_glue.npz"))


def test_alpha_composite_equals_reference_peel2mask():
    """deftet_amd.render.alpha_composite == peel2mask (5_rendereq/deftetrneder.py:31-64) on the reference-generated
    fixture (opacities at 0 and 1 included: the clamp; with and without depth layers)."""
    import torch
    from deftet_amd.render import alpha_composite
    g = _render_glue()
    ims, dep = torch.from_numpy(g["peel_ims"]), torch.from_numpy(g["peel_depth"])
    c, v, d = alpha_composite(ims, dep)
    np.testing.assert_allclose(c.numpy(), g["peel_color"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(v.numpy(), g["peel_vis"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(d.numpy(), g["peel_dep"], rtol=1e-6, atol=2e-6)
    c2, v2, d2 = alpha_composite(ims)
    assert d2 is None
    np.testing.assert_allclose(c2.numpy(), g["peel_color_nodepth"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(v2.numpy(), g["peel_vis_nodepth"], rtol=1e-6, atol=1e-6)


def test_face_attributes_and_perspective_equal_the_reference():
    """face_attributes == vertex2face (4_render/vertex2face.py:12-28, exact: a gather); perspective == cameraop.perspective
    (3_model/cameraop.py:19-33)."""
    import torch
    from deftet_amd.render import face_attributes, perspective
    g = _render_glue()
    out = face_attributes(torch.from_numpy(g["v2f_features"]), torch.from_numpy(g["v2f_faces"]))
    assert np.array_equal(out.numpy(), g["v2f_out"])
    cam, xy = perspective(torch.from_numpy(g["persp_points"]),
                          (torch.from_numpy(g["persp_rot"]), torch.from_numpy(g["persp_pos"]), torch.from_numpy(g["persp_proj"])))
    np.testing.assert_allclose(cam.numpy(), g["persp_cam"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(xy.numpy(), g["persp_xy"], rtol=1e-5, atol=1e-6)


def test_render_mesh_color_prepares_the_rasterizer_inputs_like_the_reference_call_site():
    """render_mesh_color around a rasterizer stub: the five arguments it hands the rasterizer equal the ones the
    reference's rendermeshcolor (5_rendereq/deftetrneder.py:67-113) handed ITS rasterizer, and the composited outputs for
    the same returned layers are equal — with and without the depth channel."""
    import torch
    from deftet_amd.render import render_mesh_color
    g = _render_glue()
    t = lambda k: torch.from_numpy(g[k])                     # noqa: E731
    for tag, depth in (("d", True), ("n", False)):
        seen = {}

        def stub(xy, rngs, z, img, feat):
            seen["args"] = (xy, rngs, z, img, feat)
            return t("rmc_%s_layers" % tag), None

        feat = t("rmc_feat") if depth else t("rmc_feat")[:, :, 1:]
        col, msk, dp = render_mesh_color(t("rmc_pix"), t("rmc_ranges"), t("rmc_points3d"), t("rmc_points2d"), feat, t("rmc_faces"),
                                         depth=depth, rasterizer=stub)
        xy, rngs, z, img, ff = seen["args"]
        assert np.array_equal(xy.numpy(), g["rmc_pix"]) and np.array_equal(rngs.numpy(), g["rmc_ranges"])
        assert np.array_equal(z.numpy(), g["rmc_%s_arg_z" % tag]) and np.array_equal(img.numpy(), g["rmc_%s_arg_img" % tag])
        np.testing.assert_allclose(ff.numpy(), g["rmc_%s_arg_feat" % tag], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(col.numpy(), g["rmc_%s_color" % tag], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(msk.numpy(), g["rmc_%s_mask" % tag], rtol=1e-6, atol=1e-6)
        if depth:
            np.testing.assert_allclose(dp.numpy(), g["rmc_d_depth"], rtol=1e-6, atol=2e-6)
        else:
            assert dp is None


def test_fma_contraction_flips_no_decision_on_a_configs2_sample(oracle):
    """the reference's real CUDA build ran with nvcc's default contraction (a * b + c fused), the parity
    oracle follows the source text without it.  The second oracle build (-mfma -ffp-contract=fast, same source) counts how
    many decisions the contraction flips on BASELINE configs[2]'s own data: shape 0 of the res-70 jittered grid against a
    bounded sample of its uniform queries.  (tools/fma_flip_count.py runs all 100,000 queries of several shapes on a box
    with more cores; DESIGN.md section 2 quotes its count.)  A flip needs a query within a few ulps of a face plane: of the
    order of 1e-5 of the queries, so the assertion is a generous bound, not the figure."""
    from tests import cases
    tet, pts = cases.jittered(70, 100_000, 1)
    sample = np.ascontiguousarray(pts[:, :1500])
    plain = oracle.point_in_tet(tet, sample, omp=True)
    try:
        fused = oracle.point_in_tet_contracted(tet, sample)
    except oracle.FmaOracleUnavailable as e:                 # a host without -mfma / FMA: the diagnostic is skipped, the oracle is not
        pytest.skip(str(e))
    flips = int((plain != fused).sum())
    assert flips <= 3, flips
    # where the two builds disagree the query sits on a face: the other build's answer is a face neighbour or a miss
    assert ((plain >= 0) == (fused >= 0)).mean() > 0.998


def test_forward_composition_fixture_is_selfconsistent():
    """tests/golden/forward_composition.npz (generated by running the reference's DefTet.forward_surface_align,
    deftet.py:51-130, on the CPU with the L1 operators replaced by oracle calls): the batch terms are the per-shape terms of
    DefTet.forward (:138-184) divided by the batch size and summed (:104-110), lap_v_loss is zero, both branches see the same
    geometry terms, and the recorded random numbers have the shapes sample_surf_point_batch(., 20) draws."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "forward_composition.npz"))
    ps = g["per_shape_terms"]                                            # [B, (chamfer, analytic, normal)]
    B = ps.shape[0]
    for col, name in enumerate(("sum_chamfer_distance", "sum_analytic_distance", "sum_normal_loss")):
        want = np.float32(0)
        for i in range(B):
            want = np.float32(want + np.float32(ps[i, col]) / np.float32(B))
        assert np.allclose(g["train_" + name], want, rtol=2e-6, atol=0), name
    assert (g["train_lap_v_loss"] == 0).all() and g["train_sum_normal_loss"].shape == (1,)
    for name in ("amips_energy", "edge", "volume_variance", "center_occ", "sum_analytic_distance", "sum_normal_loss"):
        assert np.array_equal(g["train_" + name], g["infer_" + name]), name        # no randomness in these
    for i in range(B):
        F = g["train_boundary_%d" % i].shape[0]
        assert g["rand_sqrt_u_%d" % i].shape == (1, F, 20, 1) and g["rand_v_%d" % i].shape == (1, F, 20, 1)
        assert np.array_equal(g["train_boundary_%d" % i], g["infer_boundary_%d" % i])
    assert g["infer_condition"].shape == (B, 200, 1) and g["train_center_occ"].shape == (B, g["tets"].shape[0])


def test_plain_c_backward_matches_the_autograd_oracle(oracle):
    """oracle_bary_bwd_f32 (bench.py's CPU fwd+bwd leg) against fp64 torch autograd of the reference formula
    (utils/tet_utils.py:28-45), duplicates and misses included."""
    from tests import cases
    tet, pts = cases.jittered(8, 1500, 2)
    cond = oracle.point_in_tet(tet, pts)
    gw = np.random.default_rng(1).standard_normal((2, 1500, 4)).astype(np.float32)
    g = oracle.bary_bwd(tet, pts, cond, gw)
    _, g64 = oracle.point_in_tet_bwd_torch(tet, pts, cond, gw)
    assert np.abs(g - g64).max() <= 5e-6 * np.abs(g64).max()
    assert (cond < 0).any() and np.abs(g).max() > 0
