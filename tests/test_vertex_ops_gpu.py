"""N2: vertex <-> tet gather, layers/DefTet/deftet.py:65-68.
Forward bit-exact vs the oracle (= torch.gather); backward bit-exact vs the oracle's sequential
fp32 sum in ascending slot order and within fp32 round-off of torch's fp64 autograd."""
import numpy as np
import pytest
import torch

from deftet_amd import grids

pytestmark = pytest.mark.gpu


def _case(res, batch, seed, per_shape_idx=False):
    verts, tets = grids.kuhn_grid(res)
    pos = grids.jittered_positions(verts, res, batch, 0.1)
    rng = np.random.default_rng(seed)
    if per_shape_idx:                                      # the reference passes tetrahedron_bxfx4: allow different lists
        idx = np.stack([tets[rng.permutation(len(tets))] for _ in range(batch)]).astype(np.int64)
    else:
        idx = tets.astype(np.int64)
    g = rng.standard_normal((batch, len(tets), 4, 3)).astype(np.float32)
    return pos.astype(np.float32), idx, g


@pytest.mark.parametrize("res,batch,per_shape", [(6, 1, False), (10, 3, False), (10, 2, True), (40, 8, False)])
def test_gather_fwd_bwd_bit_exact(cuda, oracle, res, batch, per_shape):
    from deftet_amd import hip_ops
    pos, idx, g = _case(res, batch, 7, per_shape)
    V = pos.shape[1]
    p, i, gt = torch.from_numpy(pos).to(cuda), torch.from_numpy(idx).to(cuda), torch.from_numpy(g).to(cuda)
    out = hip_ops.tet_gather(p, i, check=True)
    assert np.array_equal(out.cpu().numpy(), oracle.tet_gather(pos, idx))
    csr = hip_ops.tet_vertex_csr(i, V)
    off, slots = csr[0].cpu().numpy(), csr[1].cpu().numpy()
    assert off[0] == 0 and off[-1] == slots.size and (np.diff(off) >= 0).all()
    gp = hip_ops.tet_gather_bwd(gt, csr, V)
    want = oracle.tet_gather_bwd(g, idx, V)
    assert np.array_equal(gp.cpu().numpy(), want)                   # same summation order -> same bits
    assert torch.equal(gp, hip_ops.tet_gather_bwd(gt, csr, V))     # deterministic
    # accumulate into an existing gradient
    base = torch.randn_like(gp)
    acc = hip_ops.tet_gather_bwd(gt, csr, V, out=base.clone())
    assert torch.allclose(acc, base + gp, rtol=1e-6, atol=1e-6)
    # torch's own backward of the gather (fp64, CPU) agrees to fp32 round-off
    pt = torch.tensor(pos, dtype=torch.float64, requires_grad=True)
    ti = torch.from_numpy(idx)
    ti = ti[None].expand(batch, -1, -1) if ti.dim() == 2 else ti
    o = torch.gather(pt.unsqueeze(2).expand(-1, -1, 4, -1), 1, ti.unsqueeze(-1).expand(-1, -1, -1, 3))
    o.backward(torch.tensor(g, dtype=torch.float64))
    assert np.abs(pt.grad.numpy() - gp.cpu().numpy()).max() <= 2e-5 * np.abs(pt.grad.numpy()).max()


def test_gather_edge_cases(cuda, oracle):
    from deftet_amd import hip_ops
    # vertices without any tet, repeated vertices inside a tet, out-of-range indices
    pos = np.random.default_rng(1).standard_normal((2, 9, 3)).astype(np.float32)
    idx = np.array([[0, 0, 1, 2], [2, 1, 0, 0], [5, 5, 5, 5]], dtype=np.int64)
    g = np.random.default_rng(2).standard_normal((2, 3, 4, 3)).astype(np.float32)
    p, i, gt = torch.from_numpy(pos).to(cuda), torch.from_numpy(idx).to(cuda), torch.from_numpy(g).to(cuda)
    assert np.array_equal(hip_ops.tet_gather(p, i).cpu().numpy(), oracle.tet_gather(pos, idx))
    csr = hip_ops.tet_vertex_csr(i, 9)
    gp = hip_ops.tet_gather_bwd(gt, csr, 9).cpu().numpy()
    assert np.array_equal(gp, oracle.tet_gather_bwd(g, idx, 9))
    assert (gp[:, [3, 4, 6, 7, 8]] == 0).all()
    bad = torch.tensor([[0, 1, 2, 9]], device=cuda)
    with pytest.raises(RuntimeError):
        hip_ops.tet_gather(p, bad, check=True)
    assert torch.isnan(hip_ops.tet_gather(p, bad)[:, 0, 3]).all()
    with pytest.raises(RuntimeError):
        hip_ops.tet_vertex_csr(bad, 9)
    # empty topology
    e = torch.zeros(0, 4, dtype=torch.int64, device=cuda)
    assert hip_ops.tet_gather(p, e).shape == (2, 0, 4, 3)
    csr0 = hip_ops.tet_vertex_csr(e, 9)
    assert (hip_ops.tet_gather_bwd(torch.zeros(2, 0, 4, 3, device=cuda), csr0, 9) == 0).all()


def test_module_autograd_chain(cuda, oracle):
    """DefTet.gather_tet_pos -> point_in_tet weights -> loss: the gradient reaches the vertices and
    matches fp64 torch autograd of the reference expressions (gather + bary_centric_tet)."""
    from deftet_amd.layers.DefTet.deftet import DefTet
    from deftet_amd.layers.DefTet.check_condition_tetrahedron_base.utils import point_in_tet_bary
    verts, tets = grids.kuhn_grid(8)
    pos = grids.jittered_positions(verts, 8, 2, 0.1).astype(np.float32)
    pts = grids.random_queries(2, 500)
    p = torch.from_numpy(pos).to(cuda).requires_grad_(True)
    idx = torch.from_numpy(tets.astype(np.int64)).to(cuda)[None].expand(2, -1, -1).contiguous()
    q = torch.from_numpy(pts).to(cuda)
    m = DefTet(device=cuda)
    tet = m.gather_tet_pos(p, idx)
    cond, w = point_in_tet_bary(tet, q)
    gw = torch.randn(w.shape, device=cuda, generator=torch.Generator(device=cuda).manual_seed(5))
    (w * gw).sum().backward()
    # fp64 reference on CPU
    pc = torch.tensor(pos, dtype=torch.float64, requires_grad=True)
    tc = torch.gather(pc.unsqueeze(2).expand(-1, -1, 4, -1), 1, idx.cpu().unsqueeze(-1).expand(-1, -1, -1, 3))
    c = cond.cpu()[..., 0]
    hit = c >= 0
    sel = torch.gather(tc, 1, c.clamp(min=0).long()[:, :, None, None].expand(-1, -1, 4, 3))
    pq = torch.tensor(pts, dtype=torch.float64)
    wc = torch.stack(oracle.bary_torch(sel[:, :, 0], sel[:, :, 1], sel[:, :, 2], sel[:, :, 3], pq), dim=-1) * hit[..., None]
    (wc * gw.cpu().double()).sum().backward()
    from tests.tol import check_close
    check_close("N2 module chain grad_pos (gather + A1b backward), res8 vs fp64 autograd", p.grad, pc.grad, 5e-7, elem_rel=1e-4)


# --------------------------------------------------------------------------------------------------------------------
# A1b backward composed with the gather's backward: deftet_point_in_tet_bwd_to_vertices_f32 (round 6)
def _two_call(hip_ops, t, q, cond, gw, go, hits, csr, V, want_pts):
    out = hip_ops.point_in_tet_bwd(t, q, cond, gw, want_grad_pts=want_pts, grad_occ=go, hits=hits)
    return (hip_ops.tet_gather_bwd(out[0], csr, V),) + tuple(out[1:])


@pytest.mark.parametrize("res,n_query,batch,with_hits", [(8, 500, 2, True), (8, 500, 2, False), (6, 3000, 1, True),   # sparse / lists / dense (> 2 per tet)
                                                         (10, 4000, 3, True), (40, 50000, 8, True)])
def test_bwd_to_vertices_equals_two_call_form(cuda, res, n_query, batch, with_hits):
    """grad_pos of the fused call == tet_gather_bwd(point_in_tet_bwd(...)) bit for bit (the same additions in the same order),
    grad_pts / grad_pred likewise; on the record path (compacted rows + mask), the list path and the dense case."""
    from deftet_amd import hip_ops
    verts, tets = grids.kuhn_grid(res)
    pos = grids.jittered_positions(verts, res, batch, 0.1).astype(np.float32)
    V, T = pos.shape[1], len(tets)
    p = torch.from_numpy(pos).to(cuda)
    idx = torch.from_numpy(tets.astype(np.int64)).to(cuda)
    q = torch.from_numpy(grids.random_queries(batch, n_query)).to(cuda)
    t = hip_ops.tet_gather(p, idx)
    csr = hip_ops.tet_vertex_csr(idx, V)
    gen = torch.Generator(device=cuda).manual_seed(11)
    pred = torch.rand(batch, T, device=cuda, generator=gen)
    out = hip_ops.point_in_tet(t, q, want_bary=True, pred_bxt=pred, want_hits=with_hits)
    cond, hits = out[0], (out[3] if with_hits else None)
    gw = torch.randn(batch, n_query, 4, device=cuda, generator=gen)
    go = torch.randn(batch, n_query, device=cuda, generator=gen)
    records = with_hits and n_query <= 2 * T                          # the path that is bit-reproducible (deftet_hip.h, A1b)
    for want_pts in (False, True):
        for occ in (go, None):
            want = _two_call(hip_ops, t, q, cond, gw, occ, hits, csr, V, want_pts)
            got = hip_ops.point_in_tet_bwd_to_vertices(t, q, cond, gw, csr, V, want_grad_pts=want_pts, grad_occ=occ, hits=hits)
            assert len(got) == len(want)
            for a, b in zip(got, want):
                assert (a is None) == (b is None)
                if a is None:
                    continue
                if records:
                    assert bool((a == b).all()), (res, n_query, want_pts, occ is None, (a - b).abs().max().item())
                else:                                                  # per-tet lists threaded with atomics: the order of a tet's hits varies from call to call
                    assert (a - b).abs().max() <= 1e-5 * b.abs().max(), (res, n_query, want_pts, occ is None)
    assert want[0].abs().max() > 0
    # accumulate into an existing gradient (grad_pos and grad_pred)
    base = torch.randn(batch, V, 3, device=cuda, generator=gen)
    acc = hip_ops.point_in_tet_bwd_to_vertices(t, q, cond, gw, csr, V, grad_occ=go, hits=hits, out=base.clone())
    ref = _two_call(hip_ops, t, q, cond, gw, go, hits, csr, V, False)
    assert (acc[0] - (base + ref[0])).abs().max() <= 1e-5 * max(ref[0].abs().max().item(), 1.0)
    assert (acc[2] - ref[2]).abs().max() <= (0.0 if records else 1e-5 * ref[2].abs().max().item())


def test_bwd_to_vertices_per_shape_index_lists(cuda):
    """the reference hands every shape its own tetrahedron_bxfx4 (layers/DefTet/deftet.py:65-68): a CSR per shape (idx_batch == B),
    each shape's list in another order — the masked gather walks per-shape incidence lists and per-shape mask words"""
    from deftet_amd import hip_ops
    res, batch, n_query = 8, 3, 700
    verts, tets = grids.kuhn_grid(res)
    pos = grids.jittered_positions(verts, res, batch, 0.1).astype(np.float32)
    V, T = pos.shape[1], len(tets)
    rng = np.random.default_rng(3)
    idx_np = np.stack([tets[rng.permutation(T)] for _ in range(batch)]).astype(np.int64)
    p, idx = torch.from_numpy(pos).to(cuda), torch.from_numpy(idx_np).to(cuda)
    q = torch.from_numpy(grids.random_queries(batch, n_query)).to(cuda)
    t = hip_ops.tet_gather(p, idx)
    csr = hip_ops.tet_vertex_csr(idx, V)
    assert csr[2] == batch
    gen = torch.Generator(device=cuda).manual_seed(12)
    pred = torch.rand(batch, T, device=cuda, generator=gen)
    cond, w, occ, hits = hip_ops.point_in_tet(t, q, want_bary=True, pred_bxt=pred, want_hits=True)
    gw = torch.randn(batch, n_query, 4, device=cuda, generator=gen)
    go = torch.randn(batch, n_query, device=cuda, generator=gen)
    want = _two_call(hip_ops, t, q, cond, gw, go, hits, csr, V, True)
    got = hip_ops.point_in_tet_bwd_to_vertices(t, q, cond, gw, csr, V, want_grad_pts=True, grad_occ=go, hits=hits)
    for a, b in zip(got, want):
        assert bool((a == b).all())
    assert want[0].abs().max() > 0


def test_bwd_to_vertices_adversarial_and_empty(cuda):
    """overflowing / irregular tets (records marked, hits carried by the uncovered list), per-shape index lists, and the empty
    cases: the fused call still equals the two-call form"""
    from deftet_amd import hip_ops
    from tests import cases
    tet, pts = cases.adversarial(3)
    tet = np.concatenate([tet, tet], 1)                                # twice the soup: Q <= 2 T keeps the backward on the (bit-reproducible)
    B, T = tet.shape[0], tet.shape[1]                                  # record path; the copies accept what the originals win
    assert pts.shape[1] <= 2 * T
    # the adversarial tet soup as a topology: every tet has four vertices of its own
    pos = tet.reshape(B, T * 4, 3).copy()
    idx = np.arange(T * 4, dtype=np.int64).reshape(T, 4)
    p, i, q = torch.from_numpy(pos).to(cuda), torch.from_numpy(idx).to(cuda), torch.from_numpy(pts).to(cuda)
    V = T * 4
    t = hip_ops.tet_gather(p, i)
    csr = hip_ops.tet_vertex_csr(i, V)
    gen = torch.Generator(device=cuda).manual_seed(4)
    pred = torch.rand(B, T, device=cuda, generator=gen)
    cond, w, occ, hits = hip_ops.point_in_tet(t, q, want_bary=True, pred_bxt=pred, want_hits=True)
    rec = hits[: B * T * 2].view(B, T, 2)
    assert (rec[..., 1] == -2).any()                                   # some records are marked overflowed / irregular
    gw = torch.randn(B, q.shape[1], 4, device=cuda, generator=gen)
    go = torch.randn(B, q.shape[1], device=cuda, generator=gen)
    want = _two_call(hip_ops, t, q, cond, gw, go, hits, csr, V, True)
    got = hip_ops.point_in_tet_bwd_to_vertices(t, q, cond, gw, csr, V, want_grad_pts=True, grad_occ=go, hits=hits)
    for a, b in zip(got, want):
        f = torch.isfinite(a) & torch.isfinite(b)
        assert bool((a[f] == b[f]).all()) and bool((torch.isfinite(a) == torch.isfinite(b)).all())
    # no tets / no queries
    e_csr = hip_ops.tet_vertex_csr(torch.zeros(0, 4, dtype=torch.int64, device=cuda), 9)
    z = hip_ops.point_in_tet_bwd_to_vertices(torch.zeros(2, 0, 4, 3, device=cuda), q[:1].expand(2, -1, -1).contiguous(),
                                             torch.full((2, q.shape[1], 1), -1.0, device=cuda), gw[:1].expand(2, -1, -1).contiguous(), e_csr, 9)
    assert z[0].shape == (2, 9, 3) and (z[0] == 0).all()
    z = hip_ops.point_in_tet_bwd_to_vertices(t, torch.zeros(B, 0, 3, device=cuda), torch.zeros(B, 0, 1, device=cuda),
                                             torch.zeros(B, 0, 4, device=cuda), csr, V)
    assert (z[0] == 0).all()


def test_occupancy_query_autograd_to_vertices(cuda, oracle):
    """DefTet.occupancy_query: vertices -> (gather) -> index + weights + occ; the gradient reaches vertice_pos through the fused
    backward and matches (i) the chain gather_tet_pos -> point_in_tet_occ bit for bit, (ii) fp64 autograd of the reference
    expressions (torch.gather, layers/DefTet/deftet.py:65-68, + bary_centric_tet + paste_occ)."""
    from deftet_amd.layers.DefTet.deftet import DefTet
    from deftet_amd.layers.DefTet.check_condition_tetrahedron_base.utils import point_in_tet_occ
    verts, tets = grids.kuhn_grid(8)
    pos = grids.jittered_positions(verts, 8, 2, 0.1).astype(np.float32)
    pts = grids.random_queries(2, 500)
    idx = torch.from_numpy(tets.astype(np.int64)).to(cuda)[None].expand(2, -1, -1).contiguous()
    q = torch.from_numpy(pts).to(cuda)
    gen = torch.Generator(device=cuda).manual_seed(5)
    pred0 = torch.rand(2, len(tets), device=cuda, generator=gen)
    m = DefTet(device=cuda)
    gw = torch.randn(2, 500, 4, device=cuda, generator=gen)
    go = torch.randn(2, 500, device=cuda, generator=gen)
    grads = []
    for fused in (True, False, "given"):
        p = torch.from_numpy(pos).to(cuda).requires_grad_(True)
        pred = pred0.clone().requires_grad_(True)
        if fused is True:
            cond, w, occ = m.occupancy_query(p, idx, q, pred)
        elif fused == "given":                                         # the caller gathered already: values reused, gradient to p
            cond, w, occ = m.occupancy_query(p, idx, q, pred, tet_bxfx4x3=m.gather_tet_pos(p, idx))
        else:
            cond, w, occ = point_in_tet_occ(m.gather_tet_pos(p, idx), q, pred)
        ((w * gw).sum() + (occ * go).sum()).backward()
        grads.append((p.grad.clone(), pred.grad.clone(), cond.clone()))
    for g in grads[1:]:
        assert torch.equal(grads[0][2], g[2])
        assert bool((grads[0][0] == g[0]).all()) and bool((grads[0][1] == g[1]).all())
    # fp64 reference on CPU
    pc = torch.tensor(pos, dtype=torch.float64, requires_grad=True)
    tc = torch.gather(pc.unsqueeze(2).expand(-1, -1, 4, -1), 1, idx.cpu().unsqueeze(-1).expand(-1, -1, -1, 3))
    c = grads[0][2].cpu()[..., 0]
    hit = c >= 0
    sel = torch.gather(tc, 1, c.clamp(min=0).long()[:, :, None, None].expand(-1, -1, 4, 3))
    pq = torch.tensor(pts, dtype=torch.float64)
    wc = torch.stack(oracle.bary_torch(sel[:, :, 0], sel[:, :, 1], sel[:, :, 2], sel[:, :, 3], pq), dim=-1) * hit[..., None]
    (wc * gw.cpu().double()).sum().backward()
    from tests.tol import check_close
    check_close("A1b o N2 fused backward grad_pos (DefTet.occupancy_query), res8 vs fp64 autograd", grads[0][0], pc.grad, 5e-7, elem_rel=1e-4)
