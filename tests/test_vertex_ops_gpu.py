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
