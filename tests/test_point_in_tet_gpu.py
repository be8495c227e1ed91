"""GPU parity tests for A1 / A1b: HIP path (through the C ABI) vs the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from tests import cases
from tests.tol import check_close

pytestmark = pytest.mark.gpu


def _run(tet, pts, dev, algo=0, bary=False):
    from deftet_amd import hip_ops
    t = torch.from_numpy(tet).to(dev)
    p = torch.from_numpy(pts).to(dev)
    out = hip_ops.point_in_tet(t, p, want_bary=bary, algo=algo)
    torch.cuda.synchronize()
    if bary:
        return out[0].cpu().numpy(), out[1].cpu().numpy()
    return out.cpu().numpy()


@pytest.mark.parametrize("algo", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("res,nq,batch", [(4, 257, 1), (8, 3000, 3), (12, 5000, 2)])
def test_index_bit_exact_jittered(cuda, oracle, algo, res, nq, batch):
    tet, pts = cases.jittered(res, nq, batch)
    want = oracle.point_in_tet(tet, pts)
    got = _run(tet, pts, cuda, algo)
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.array_equal(got, want)
    assert 0.05 < (want < 0).mean() < 0.25          # the 13.6 % miss band of SURVEY 3.2


@pytest.mark.parametrize("algo", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_index_bit_exact_adversarial(cuda, oracle, algo, seed):
    tet, pts = cases.adversarial(seed)
    want = oracle.point_in_tet(tet, pts)
    got = _run(tet, pts, cuda, algo)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("scale,offset", [(1e-5, (0, 0, 0)), (1e4, (0, 0, 0)), (1.0, (1000.0, -2000.0, 500.0)),
                                          (1e-3, (7.0, 7.0, 7.0)), (3e5, (1e5, 0, 0))])
@pytest.mark.parametrize("algo", [0, 2, 3, 4, 5])
def test_index_bit_exact_scaled(cuda, oracle, scale, offset, algo):
    tet, pts = cases.scaled(scale, offset)
    want = oracle.point_in_tet(tet, pts)
    got = _run(tet, pts, cuda, algo)
    assert np.array_equal(got, want)


def test_empty_and_ragged(cuda):
    from deftet_amd import hip_ops
    tet = torch.zeros(2, 0, 4, 3, device=cuda)
    pts = torch.rand(2, 10, 3, device=cuda)
    assert (hip_ops.point_in_tet(tet, pts) == -1).all()
    tet, p = cases.jittered(4, 5, 1)
    out = hip_ops.point_in_tet(torch.from_numpy(tet).to(cuda), torch.zeros(1, 0, 3, device=cuda))
    assert out.shape == (1, 0, 1)
    # no tets: the backward still defines grad_pts (zeros), not uninitialised memory
    tet0 = torch.zeros(2, 0, 4, 3, device=cuda)
    pts = torch.rand(2, 10, 3, device=cuda)
    cond = hip_ops.point_in_tet(tet0, pts)
    g_tet, g_pts = hip_ops.point_in_tet_bwd(tet0, pts, cond, torch.ones(2, 10, 4, device=cuda), want_grad_pts=True)
    assert g_tet.shape == (2, 0, 4, 3) and (g_pts == 0).all()


def test_binned_equals_brute_res40(cuda):
    """size-independent property at a BASELINE size: two independent GPU algorithms agree."""
    tet, pts = cases.jittered(40, 50000, 2)
    a = _run(tet, pts, cuda, 0)
    b = _run(tet, pts, cuda, 1)
    assert np.array_equal(a, b)
    for algo in (2, 3, 4, 5):
        assert np.array_equal(a, _run(tet, pts, cuda, algo)), algo
    assert 0.10 < (a < 0).mean() < 0.17


def test_weights_and_backward(cuda, oracle):
    from deftet_amd.layers.DefTet.check_condition_tetrahedron_base.utils import point_in_tet_bary
    tet, pts = cases.jittered(8, 4000, 2)
    t = torch.from_numpy(tet).to(cuda).requires_grad_(True)
    p = torch.from_numpy(pts).to(cuda).requires_grad_(True)
    cond, w = point_in_tet_bary(t, p)
    want_c = oracle.point_in_tet(tet, pts)
    assert np.array_equal(cond.detach().cpu().numpy(), want_c)
    gw = torch.from_numpy(np.random.default_rng(4000).standard_normal((2, 4000, 4)).astype(np.float32)).to(cuda)
    (w * gw).sum().backward()
    w64, gt64 = oracle.point_in_tet_bwd_torch(tet, pts, want_c, gw.cpu().numpy())
    hit = want_c[..., 0] >= 0
    # tolerance: 1e-5 relative (north_star), measured against the tensor's scale
    wn = w.detach().cpu().numpy()
    assert np.abs(wn - w64)[hit].max() <= 1e-5 * max(1.0, np.abs(w64).max())
    assert (wn[~hit] == 0).all()
    assert np.allclose(wn[hit].sum(-1), 1.0, atol=1e-5)
    g = t.grad.cpu().numpy()
    scale = np.abs(gt64).max()
    assert np.abs(g - gt64).max() <= 1e-5 * scale * 8     # up to ~8 atomically-summed contributions per tet
    assert p.grad is not None and torch.isfinite(p.grad).all()


def test_paste_occ(cuda):
    from deftet_amd.layers.DefTet.check_condition_tetrahedron_base.utils import paste_occ
    g = torch.Generator().manual_seed(0)
    pred = torch.rand(3, 50, generator=g).to(cuda).requires_grad_(True)
    cond = torch.randint(-1, 50, (3, 200, 1), generator=g).float().to(cuda)
    c2 = cond.clone()
    out = paste_occ(pred, c2)
    ref_c = cond.clone(); ref_c[ref_c < 0] = 0
    ref = torch.gather(pred, 1, ref_c.long().squeeze(-1))
    assert torch.equal(out, ref) and torch.equal(c2, ref_c)
    go = torch.rand(3, 200, generator=g).to(cuda)
    out.backward(go)
    pr = pred.detach().clone().requires_grad_(True)
    torch.gather(pr, 1, ref_c.long().squeeze(-1)).backward(go)
    assert torch.allclose(pred.grad, pr.grad, rtol=1e-5, atol=1e-6)


def test_rejects_cpu_tensors():
    from deftet_amd import hip_ops, _lib
    with pytest.raises(_lib.DefTetHipError):
        hip_ops.point_in_tet(torch.zeros(1, 1, 4, 3), torch.zeros(1, 1, 3))


def test_backward_atomic_fallback_matches_gather(cuda, oracle):
    """deftet_point_in_tet_bwd_f32 with workspace=NULL (float-atomic scatter) and with a
    workspace (linked-list gather) must agree; accumulate=1 adds to the existing content."""
    import ctypes as C
    from deftet_amd import _lib, hip_ops
    lib = _lib.load()
    tet, pts = cases.jittered(8, 3000, 2)
    t = torch.from_numpy(tet).to(cuda)
    p = torch.from_numpy(pts).to(cuda)
    cond = hip_ops.point_in_tet(t, p)
    gw = torch.randn(2, 3000, 4, device=cuda, generator=torch.Generator(device=cuda).manual_seed(1))
    g_gather, gp = hip_ops.point_in_tet_bwd(t, p, cond, gw, want_grad_pts=True)
    g_atomic = torch.full_like(t, 7.0)          # must be overwritten when accumulate == 0
    st = _lib.current_stream(cuda)
    _lib.check(lib.deftet_point_in_tet_bwd_f32(_lib.ptr(t), _lib.ptr(p), _lib.ptr(cond), _lib.ptr(gw), _lib.ptr(g_atomic),
                                               None, None, None, None, 2, t.shape[1], 3000, 0, None, 0, st), "bwd atomic")
    torch.cuda.synchronize()
    scale = g_gather.abs().max().item()
    assert (g_gather - g_atomic).abs().max().item() <= 1e-5 * scale
    acc = torch.ones_like(t)
    ws = _lib.workspace(cuda, lib.deftet_point_in_tet_bwd_workspace_bytes(2, t.shape[1], 3000))
    _lib.check(lib.deftet_point_in_tet_bwd_f32(_lib.ptr(t), _lib.ptr(p), _lib.ptr(cond), _lib.ptr(gw), _lib.ptr(acc),
                                               None, None, None, None, 2, t.shape[1], 3000, 1, _lib.ptr(ws), ws.numel(), st), "bwd acc")
    torch.cuda.synchronize()
    assert (acc - 1.0 - g_gather).abs().max().item() <= 1e-5 * scale
    miss = cond[..., 0] < 0
    assert (gp[miss] == 0).all() and gp[~miss].abs().sum() > 0


def test_rowdot(cuda):
    from deftet_amd import hip_ops
    g = torch.Generator(device=cuda).manual_seed(3)
    for shape in [(8, 100000, 4), (3, 1001), (2, 7, 3)]:
        a = torch.randn(*shape, device=cuda, generator=g)
        b = torch.randn(*shape, device=cuda, generator=g)
        want = (a.double() * b.double()).flatten(1).sum(1)
        got = hip_ops.rowdot(a, b)
        assert torch.allclose(got.double(), want, rtol=1e-5, atol=1e-3)
        assert torch.equal(got, hip_ops.rowdot(a, b))            # deterministic
        assert torch.allclose(hip_ops.rowdot(a).double(), a.double().flatten(1).sum(1), rtol=1e-5, atol=1e-3)
        a2 = torch.randn(shape[0], 1237, device=cuda, generator=g)                # second pair of another width
        b2 = torch.randn(shape[0], 1237, device=cuda, generator=g)
        want2 = want + (a2.double() * b2.double()).sum(1)
        assert torch.allclose(hip_ops.rowdot(a, b, a2, b2).double(), want2, rtol=1e-5, atol=1e-3)
        assert torch.allclose(hip_ops.rowdot(a, b, a2).double(), want + a2.double().sum(1), rtol=1e-5, atol=1e-3)


def test_fused_occ_op_matches_separate_ops(cuda, oracle):
    """point_in_tet_occ == check_condition + bary + paste_occ (values and both gradients), with
    the atomic fallback of the fused backward checked too."""
    from deftet_amd import _lib, hip_ops
    from deftet_amd.layers.DefTet.check_condition_tetrahedron_base.utils import (paste_occ, point_in_tet_bary,
                                                                                point_in_tet_occ)
    tet, pts = cases.jittered(8, 5000, 3)
    gen = torch.Generator(device=cuda).manual_seed(7)
    T = tet.shape[1]
    pred0 = torch.rand(3, T, device=cuda, generator=gen)
    gw = torch.randn(3, 5000, 4, device=cuda, generator=gen)
    go = torch.randn(3, 5000, device=cuda, generator=gen)

    t1 = torch.from_numpy(tet).to(cuda).requires_grad_(True)
    p1 = pred0.clone().requires_grad_(True)
    cond, w, occ = point_in_tet_occ(t1, torch.from_numpy(pts).to(cuda), p1)
    ((w * gw).sum() + (occ * go).sum()).backward()

    t2 = torch.from_numpy(tet).to(cuda).requires_grad_(True)
    p2 = pred0.clone().requires_grad_(True)
    cond2, w2 = point_in_tet_bary(t2, torch.from_numpy(pts).to(cuda))
    occ2 = paste_occ(p2, cond2.clone())
    ((w2 * gw).sum() + (occ2 * go).sum()).backward()

    assert torch.equal(cond, cond2) and (cond < 0).any()                 # fused op keeps the -1s
    assert np.array_equal(cond.cpu().numpy(), oracle.point_in_tet(tet, pts))
    assert torch.equal(w, w2) and torch.equal(occ, occ2)
    assert (t1.grad - t2.grad).abs().max() <= 1e-5 * t2.grad.abs().max()
    assert (p1.grad - p2.grad).abs().max() <= 1e-5 * p2.grad.abs().max()
    # misses paste from (and send their gradient to) tet 0
    miss = cond[..., 0] < 0
    assert torch.equal(occ[miss], pred0[:, 0:1].expand(-1, 5000)[miss])
    # atomic fallback (no workspace) of the fused backward
    lib = _lib.load()
    g_tet = torch.empty_like(t1)
    g_pred = torch.empty(3, T, device=cuda)
    tt, pp = t1.detach().contiguous(), torch.from_numpy(pts).to(cuda)
    _lib.check(lib.deftet_point_in_tet_bwd_f32(_lib.ptr(tt), _lib.ptr(pp), _lib.ptr(cond), _lib.ptr(gw), _lib.ptr(g_tet), None,
                                               _lib.ptr(go), _lib.ptr(g_pred), None, 3, T, 5000, 0, None, 0, _lib.current_stream(cuda)),
               "fused bwd atomic")
    torch.cuda.synchronize()
    assert (g_pred - p1.grad).abs().max() <= 1e-5 * p1.grad.abs().max()
    assert (g_tet - t1.grad).abs().max() <= 1e-5 * t1.grad.abs().max()


@pytest.mark.parametrize("algo", [0, 2, 3, 4, 5])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_backward_hit_records_adversarial(cuda, oracle, seed, algo):
    """the three backward paths (hit records / linked lists / atomics) agree, including tets that
    swallow many queries (record overflow), irregular tets and NaN / huge queries"""
    from deftet_amd import _lib, hip_ops
    tet, pts = cases.adversarial(seed)
    pts = np.nan_to_num(pts, nan=0.123, posinf=0.3, neginf=-0.3)        # finite coordinates: gradients stay finite
    pts[0, 5] = 3.0e6                                                  # one huge (irregular) query stays
    B, T, Q = 1, tet.shape[1], pts.shape[1]
    finite_t = np.isfinite(tet).all(axis=(2, 3))[0]
    tet = np.ascontiguousarray(tet[:, finite_t])                       # drop NaN/Inf tets (their weights are NaN by definition)
    T = tet.shape[1]
    t, p = torch.from_numpy(tet).to(cuda), torch.from_numpy(pts).to(cuda)
    gen = torch.Generator(device=cuda).manual_seed(seed)
    pred = torch.rand(B, T, device=cuda, generator=gen)
    cond, w, occ, hits = hip_ops.point_in_tet(t, p, want_bary=True, pred_bxt=pred, want_hits=True, algo=algo)
    assert np.array_equal(cond.cpu().numpy(), oracle.point_in_tet(tet, pts))
    h2 = hits[: B * T * 2].view(B, T, 2)                               # the 8-byte records (round 6)
    assert (h2[..., 1] == -2).any()                                    # some tet overflowed / is irregular
    gw = torch.randn(B, Q, 4, device=cuda, generator=gen)
    go = torch.randn(B, Q, device=cuda, generator=gen)
    a = hip_ops.point_in_tet_bwd(t, p, cond, gw, want_grad_pts=True, grad_occ=go, hits=hits)
    b = hip_ops.point_in_tet_bwd(t, p, cond, gw, want_grad_pts=True, grad_occ=go)
    fin = torch.isfinite(b[0]) & torch.isfinite(a[0])
    assert fin.float().mean() > 0.9
    scale = b[0][fin].abs().max()
    assert ((a[0] - b[0])[fin]).abs().max() <= 2e-5 * scale
    assert torch.equal(torch.isfinite(a[0]), torch.isfinite(b[0]))
    assert (a[2] - b[2]).abs().max() <= 1e-4 * b[2].abs().max()       # grad_pred (thousands of hits on the swallowing tets)
    fp = torch.isfinite(b[1])
    assert ((a[1] - b[1])[fp]).abs().max() <= 2e-5 * b[1][fp].abs().max()


def test_binned_equals_brute_res100(cuda):
    """BASELINE configs[3] shape size (res=100: T=750,000, Q=200,000): the binned path against the
    independent brute-force kernel (1.5e11 pair tests), plus round-trip properties of the outputs."""
    from deftet_amd import hip_ops
    tet, pts = cases.jittered(100, 200000, 1)
    t, p = torch.from_numpy(tet).to(cuda), torch.from_numpy(pts).to(cuda)
    cond, w = hip_ops.point_in_tet(t, p, want_bary=True)
    brute = hip_ops.point_in_tet(t, p, algo=1)
    assert torch.equal(cond, brute)
    hit = cond[..., 0] >= 0
    assert 0.10 < (~hit).float().mean().item() < 0.17
    # weights reconstruct the query from the hit tet's vertices (sum w_i v_i = p) and are a partition of unity
    idx = cond[..., 0].clamp(min=0).long()
    verts = torch.gather(t, 1, idx[:, :, None, None].expand(-1, -1, 4, 3))
    rec = (w[..., None] * verts).sum(2)
    assert (rec - p)[hit].abs().max().item() < 2e-6
    assert (w[hit].sum(-1) - 1).abs().max().item() < 1e-5 and w[hit].min().item() > -1e-5
    # queries outside the grid never hit
    outside = (p.abs() > 0.5).any(-1)
    assert not (hit & outside).any()


@pytest.mark.parametrize("B,T,Q", [(1, 1, 1), (2, 3, 2), (1, 63, 64), (5, 64, 65), (1, 65, 2047), (2, 200, 2048), (1, 1000, 2049),
                                   (9, 37, 4100), (1, 5000, 6000)])
@pytest.mark.parametrize("pattern", ["uniform", "clustered", "plane"])
def test_random_soups_and_query_patterns(cuda, oracle, B, T, Q, pattern):
    """overlapping random tets (many queries accepted by several tets) x query sets that stress the
    counting sort: everything in one cell / one row, sizes around the 2048-query chunk and the
    64-lane wave; all binned variants against the oracle, hit-record backward against the list backward"""
    from deftet_amd import hip_ops
    rng = np.random.default_rng(B * 1000003 + T * 1009 + Q)
    c = rng.random((B, T, 1, 3)).astype(np.float32) - 0.5
    tet = (c + 0.35 * (rng.random((B, T, 4, 3)).astype(np.float32) - 0.5)).astype(np.float32)
    if pattern == "uniform":
        pts = (rng.random((B, Q, 3)) - 0.5).astype(np.float32)
    elif pattern == "clustered":                                       # one grid cell, a few stragglers that span the box
        pts = (0.1 + 1e-4 * rng.random((B, Q, 3))).astype(np.float32)
        pts[:, : max(1, Q // 50)] = (rng.random((B, max(1, Q // 50), 3)) - 0.5).astype(np.float32)
    else:                                                              # a plane: one (cz) slab of rows
        pts = (rng.random((B, Q, 3)) - 0.5).astype(np.float32)
        pts[..., 2] = 0.0625
    want = oracle.point_in_tet(tet, pts)
    t, p = torch.from_numpy(tet).to(cuda), torch.from_numpy(pts).to(cuda)
    for algo in (0, 2, 3, 4):
        cond = hip_ops.point_in_tet(t, p, algo=algo)
        assert np.array_equal(cond.cpu().numpy(), want), algo
    gen = torch.Generator(device=cuda).manual_seed(1)
    pred = torch.rand(B, T, device=cuda, generator=gen)
    cond, w, occ, hits = hip_ops.point_in_tet(t, p, want_bary=True, pred_bxt=pred, want_hits=True)
    assert np.array_equal(cond.cpu().numpy(), want)
    gw, go = torch.randn(B, Q, 4, device=cuda, generator=gen), torch.randn(B, Q, device=cuda, generator=gen)
    a = hip_ops.point_in_tet_bwd(t, p, cond, gw, want_grad_pts=True, grad_occ=go, hits=hits)
    b = hip_ops.point_in_tet_bwd(t, p, cond, gw, want_grad_pts=True, grad_occ=go)
    for x, y in zip(a, b):
        f = torch.isfinite(x) & torch.isfinite(y)
        assert f.float().mean() > 0.99
        assert ((x - y)[f]).abs().max() <= 1e-4 * max(y[f].abs().max().item(), 1e-30)


def test_more_queries_than_row_blocks(cuda):
    """Q beyond 256 chunks x 2048 (the counting sort then loops inside a block) and Q >> T: the
    binned path against the independent brute-force kernel"""
    from deftet_amd import hip_ops
    tet, _ = cases.jittered(14, 10, 2)
    rng = np.random.default_rng(5)
    pts = (1.05 * (rng.random((2, 600000, 3)) - 0.5)).astype(np.float32)
    t, p = torch.from_numpy(tet).to(cuda), torch.from_numpy(pts).to(cuda)
    cond, w, hits = hip_ops.point_in_tet(t, p, want_bary=True, want_hits=True)
    assert torch.equal(cond, hip_ops.point_in_tet(t, p, algo=1))
    hit = cond[..., 0] >= 0
    assert 0.8 < hit.float().mean().item() < 0.9
    assert (w[hit].sum(-1) - 1).abs().max().item() < 1e-5
    # every tet swallows ~250 queries: all records overflow, the backward runs through the uncovered list
    gw = torch.randn(2, 600000, 4, device=cuda, generator=torch.Generator(device=cuda).manual_seed(2))
    a = hip_ops.point_in_tet_bwd(t, p, cond, gw, hits=hits)[0]
    b = hip_ops.point_in_tet_bwd(t, p, cond, gw)[0]
    assert (a - b).abs().max() <= 1e-3 * b.abs().max()


def test_prepared_queries_two_streams(cuda, oracle):
    """deftet_point_in_tet_prepare_f32 + _scan_f32 == the fused call, with the query sort on another stream"""
    from deftet_amd import hip_ops
    tet, pts = cases.jittered(10, 4000, 3)
    tet2 = tet[:, ::-1].copy()                                        # another tet set for the same queries
    t, t2, p = torch.from_numpy(tet).to(cuda), torch.from_numpy(tet2).to(cuda), torch.from_numpy(pts).to(cuda)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        pq = hip_ops.prepare_queries(p, t.shape[1])
        pq2 = hip_ops.prepare_queries(p, t.shape[1], algo=2)
    pred = torch.rand(3, t.shape[1], device=cuda, generator=torch.Generator(device=cuda).manual_seed(3))
    cond, w, occ, hits = hip_ops.point_in_tet(t, p, want_bary=True, pred_bxt=pred, want_hits=True, prepared=pq)
    ref = hip_ops.point_in_tet(t, p, want_bary=True, pred_bxt=pred, want_hits=True)
    assert np.array_equal(cond.cpu().numpy(), oracle.point_in_tet(tet, pts))
    assert torch.equal(cond, ref[0]) and torch.equal(w, ref[1]) and torch.equal(occ, ref[2])
    gw = torch.randn_like(w)
    a = hip_ops.point_in_tet_bwd(t, p, cond, gw, hits=hits)[0]
    b = hip_ops.point_in_tet_bwd(t, p, ref[0], gw, hits=ref[3])[0]
    assert (a - b).abs().max() <= 1e-5 * b.abs().max()
    cond2 = hip_ops.point_in_tet(t2, p, algo=2, prepared=pq2)
    assert np.array_equal(cond2.cpu().numpy(), oracle.point_in_tet(tet2, pts))
    with pytest.raises(RuntimeError):
        hip_ops.point_in_tet(t, p, prepared=pq)                       # a prepare feeds exactly one scan
    with pytest.raises(RuntimeError):
        hip_ops.prepare_queries(p, t.shape[1], algo=1)


# ---------------------------------------------------------------------------------------------
# reference-derived pins and the BASELINE.json configurations at full size
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("algo", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("name", ["kuhn4", "kuhn8", "kuhn20", "soup", "cube40"])
def test_index_pinned_by_reference_barycentrics(cuda, name, algo):
    """HIP path vs tests/golden/pit_index_*.npz: expected index from the reference's own
    bary_centric_tet (utils/tet_utils.py:28-45) evaluated on every (tet, query) pair; queries that
    touch a tet within 1e-4 are masked (`ambiguous`)."""
    from tests.test_cpu_oracle_golden import _pit_fixture
    tet, pts, expected, ambiguous, w_ref = _pit_fixture(name)
    cond, w = _run(tet[None], pts[None], cuda, algo, bary=True) if algo != 1 else (_run(tet[None], pts[None], cuda, algo), None)
    got = cond[0, :, 0].astype(np.int64)
    clear = ~ambiguous
    assert np.array_equal(got[clear], expected[clear].astype(np.int64))
    if w is not None:
        sel = clear & (expected >= 0)
        assert np.abs(w[0][sel] - w_ref[sel]).max() <= 1e-5 * max(1.0, np.abs(w_ref[sel]).max())


def _check_outputs(t, p, cond, w):
    """size-independent properties of (index, weights): partition of unity, sum w_i v_i = p, no hit outside the grid"""
    hit = cond[..., 0] >= 0
    idx = cond[..., 0].clamp(min=0).long()
    verts = torch.gather(t, 1, idx[:, :, None, None].expand(-1, -1, 4, 3))
    rec = (w[..., None] * verts).sum(2)
    assert (rec - p)[hit].abs().max().item() < 2e-6
    assert (w[hit].sum(-1) - 1).abs().max().item() < 1e-5 and w[hit].min().item() > -1e-5
    assert (w[~hit] == 0).all()
    assert not (hit & (p.abs() > 0.5).any(-1)).any()
    return hit


def test_config0_res20_10k_b1_vs_oracle(cuda, oracle):
    """BASELINE configs[0]: res=20 grid (T=6,000), 10k queries, batch 1 — HIP (all traversal variants and the
    brute kernel) vs the full CPU oracle, plus weights/backward vs fp64 autograd of the reference formula."""
    from deftet_amd import hip_ops
    tet, pts = cases.jittered(20, 10000, 1)
    want = oracle.point_in_tet(tet, pts, omp=True)
    for algo in (0, 1, 2, 3, 4, 5):
        assert np.array_equal(_run(tet, pts, cuda, algo), want), algo
    t, p = torch.from_numpy(tet).to(cuda), torch.from_numpy(pts).to(cuda)
    cond, w, hits = hip_ops.point_in_tet(t, p, want_bary=True, want_hits=True)
    _check_outputs(t, p, cond, w)
    gw = torch.from_numpy(np.random.default_rng(4000).standard_normal((1, 10000, 4)).astype(np.float32)).to(cuda)
    g = hip_ops.point_in_tet_bwd(t, p, cond, gw, hits=hits)[0].cpu().numpy()
    w64, gt64 = oracle.point_in_tet_bwd_torch(tet, pts, want, gw.cpu().numpy())
    hit = want[..., 0] >= 0
    check_close("A1b weights, configs[0] res20 10k B1 vs fp64", w.cpu().numpy()[hit], w64[hit], 1e-6, elem_rel=4e-4)
    check_close("A1b grad_tet, configs[0] res20 10k B1 vs fp64 autograd", g, gt64, 1e-6, elem_rel=3e-4)


@pytest.mark.parametrize("res,nq,batch,sub", [(40, 50000, 8, 4000), (70, 100000, 8, 2000)])
def test_config_full_size_b8(cuda, oracle, res, nq, batch, sub):
    """BASELINE configs[1] (res=40, 50k, B=8) and configs[2] (res=70, 100k, B=8) at FULL size:
    binned == independent brute-force kernel on every query of every shape, HIP == CPU oracle (OpenMP) on
    a `sub`-query subsample of every shape, output properties, and the hit-record backward against the
    list backward."""
    from deftet_amd import hip_ops
    tet, pts = cases.jittered(res, nq, batch)
    t, p = torch.from_numpy(tet).to(cuda), torch.from_numpy(pts).to(cuda)
    gen = torch.Generator(device=cuda).manual_seed(res)
    pred = torch.rand(batch, tet.shape[1], device=cuda, generator=gen)
    cond, w, occ, hits = hip_ops.point_in_tet(t, p, want_bary=True, pred_bxt=pred, want_hits=True)
    brute = hip_ops.point_in_tet(t, p, algo=1)
    assert torch.equal(cond, brute)
    for algo in (2, 3, 4, 5):                     # the other traversal variants, same full-size input
        assert torch.equal(hip_ops.point_in_tet(t, p, algo=algo), brute), algo
    hit = _check_outputs(t, p, cond, w)
    assert 0.10 < (~hit).float().mean().item() < 0.17
    # CPU oracle on a subsample of each shape (the oracle scans all T tets per query)
    rng = np.random.default_rng(res)
    pick = np.sort(rng.choice(nq, sub, replace=False))
    want = oracle.point_in_tet(tet, np.ascontiguousarray(pts[:, pick]), omp=True)
    assert np.array_equal(cond[:, pick].cpu().numpy(), want)
    # pasted occupancy = pred[cond] with misses aliased to tet 0 (deftet.py:132-136)
    idx = cond[..., 0].clamp(min=0).long()
    assert torch.equal(occ, torch.gather(pred, 1, idx))
    # backward: hit records vs per-tet lists
    gw = torch.randn(batch, nq, 4, device=cuda, generator=gen)
    go = torch.randn(batch, nq, device=cuda, generator=gen)
    a = hip_ops.point_in_tet_bwd(t, p, cond, gw, grad_occ=go, hits=hits)
    b = hip_ops.point_in_tet_bwd(t, p, cond, gw, grad_occ=go)
    assert (a[0] - b[0]).abs().max() <= 2e-5 * b[0].abs().max()
    assert (a[2] - b[2]).abs().max() <= 1e-4 * b[2].abs().max()
    # backward and weights at FULL size against the fp64 autograd of the reference formula (utils/tet_utils.py:28-45) on the
    # GPU's own `cond` (already proven equal to the brute-force kernel above): every query of all `batch` shapes
    w64, gt64 = oracle.point_in_tet_bwd_torch(tet, pts, cond.cpu().numpy(), gw.cpu().numpy())
    hitn = hit.cpu().numpy()
    # ... and BIT-identical to the fp32 evaluation of the same formula in the torch expression's association (the oracle's
    # oracle_bary_f32, -ffp-contract=off): what the reference's own fp32 run computes for the tet `cond` names
    assert np.array_equal(w.cpu().numpy(), oracle.bary(tet, pts, cond.cpu().numpy()))
    check_close("A1b weights, res%d %dk B%d vs fp64" % (res, nq // 1000, batch), w.cpu().numpy()[hitn], w64[hitn], 1e-6, elem_rel=4e-4)
    check_close("A1b grad_tet, res%d %dk B%d vs fp64 autograd" % (res, nq // 1000, batch), a[0], gt64, 1e-6, elem_rel=3e-4)
    for algo in (2, 3, 4, 5):                              # the other traversals: their hit records drive the same backward
        c2, w2, o2, h2 = hip_ops.point_in_tet(t, p, want_bary=True, pred_bxt=pred, want_hits=True, algo=algo)
        assert torch.equal(c2, cond) and torch.equal(w2, w) and torch.equal(o2, occ)
        a2 = hip_ops.point_in_tet_bwd(t, p, c2, gw, grad_occ=go, hits=h2)
        assert (a2[0] - b[0]).abs().max() <= 2e-5 * b[0].abs().max()
        assert (a2[2] - b[2]).abs().max() <= 1e-4 * b[2].abs().max()


@pytest.mark.parametrize("res,nq", [(8, 150), (8, 1200), (20, 4000)])
def test_hit_records_equal_between_traversal_kernels(cuda, res, nq):
    """The per-tet hit records of the default traversal (published with hand-written stores) hold the same query sets
    as the ones of the exact kernel (plain compiler-generated stores); overflowed records only need to agree on the flag.
    (Round 3: a missing wait state after a 128-bit store once zeroed the first slot of 4 lanes in 16.)"""
    from deftet_amd import grids, hip_ops
    tet, pts, _, _ = grids.make_case(res, nq, 2)
    t, p = torch.from_numpy(tet).to(cuda), torch.from_numpy(pts).to(cuda)
    B, T = t.shape[0], t.shape[1]
    Q = p.shape[1]
    SPILLED = 1 << 30

    def records(algo):
        """Per tet: the sorted tuple of recorded queries, or None when the tet is marked overflowed.  Every kernel keeps up to six
        (two in the 8-byte record, four in the spill record, flag in slot 0; round 6)."""
        cond, hits = hip_ops.point_in_tet(t, p, want_hits=True, algo=algo)
        rec = hits[:2 * B * T].view(B, T, 2).cpu().numpy()
        pad = (B + 63) // 64 * 64
        off = (2 * B * T + 3 * pad + B * Q + 3) // 4 * 4
        spill = hits[off:off + 4 * B * T].view(B, T, 4).cpu().numpy()
        out = {}
        for bi in range(B):
            for ti in range(T):
                r = rec[bi, ti]
                if r[1] == -2:
                    out[bi, ti] = None
                    continue
                ids = list(r)
                if r[0] >= 0 and r[0] & SPILLED:
                    ids[0] = r[0] & ~SPILLED
                    ids += list(spill[bi, ti])
                out[bi, ti] = tuple(sorted(int(x) for x in ids if x >= 0))
        return cond, out

    cond, rec0 = records(4)
    _, rec2 = records(2)
    _, rec3 = records(3)                                               # the round-3 traversal: six slots too (four, then two), but a
    for key, ids3 in rec3.items():                                     # batch of three candidates may push it over the edge early
        ids0 = rec0[key]
        if ids3 is not None and ids0 is not None:
            assert ids0 == ids3, (key, ids0, ids3)
        elif ids3 is not None:
            assert False, (key, ids3)                                  # the default overflows at seven: so does the slab kernel
        else:
            assert ids0 is None or 5 <= len(ids0) <= 6, (key, ids0)
    n_spilled = 0
    for key, ids2 in rec2.items():
        ids0 = rec0[key]
        assert ids0 == ids2, (key, ids0, ids2)                        # the exact kernel: the same six slots, the same sets
        if ids0 is not None and len(ids0) > 2:
            n_spilled += 1
    if nq >= 4 * T // 3:
        assert n_spilled > 0                                          # the dense case really exercises the spill record
    # and every winner of a tet that is not overflowed is in that tet's record(s)
    c = cond[..., 0].long().cpu().numpy()
    for bi in range(B):
        for q in np.nonzero(c[bi] >= 0)[0]:
            ids = rec0[bi, int(c[bi, q])]
            assert ids is None or int(q) in ids


def test_config3_full_size_b8(cuda, oracle):
    """BASELINE configs[3] at one GPU's share — res=100 (T = 750,000), 200k queries, EIGHT shapes (what bench.py --config 3
    times): the binned path == the independent brute-force kernel on every query of every shape, output properties, and the
    CPU oracle on a subsample."""
    from deftet_amd import hip_ops
    tet, pts = cases.jittered(100, 200000, 8)
    t, p = torch.from_numpy(tet).to(cuda), torch.from_numpy(pts).to(cuda)
    cond, w = hip_ops.point_in_tet(t, p, want_bary=True)
    brute = hip_ops.point_in_tet(t, p, algo=1)
    assert torch.equal(cond, brute)
    hit = _check_outputs(t, p, cond, w)
    assert 0.10 < (~hit).float().mean().item() < 0.17
    pick = np.sort(np.random.default_rng(3).choice(200000, 300, replace=False))
    want = oracle.point_in_tet(tet, np.ascontiguousarray(pts[:, pick]), omp=True)
    assert np.array_equal(cond[:, pick].cpu().numpy(), want)


# ---------------------------------------------------------------------------------------------------------------------
# traversal order (round 5): the result must not depend on how the tets are numbered, nor on the permutation they are
# traversed in
# ---------------------------------------------------------------------------------------------------------------------
def test_spatial_order_is_a_permutation_and_groups_columns(cuda):
    from deftet_amd import grids, hip_ops
    tet, _, _, _ = grids.make_case(20, 10, 1)
    rng = np.random.default_rng(3)
    shuffled = np.ascontiguousarray(tet[0][rng.permutation(tet.shape[1])])
    for arr, coherent in ((tet[0], True), (shuffled, False)):
        order, breaks = hip_ops.tet_spatial_order(torch.from_numpy(arr).to(cuda), want_breaks=True)
        o = order.cpu().numpy()
        assert np.array_equal(np.sort(o), np.arange(arr.shape[0]))
        native, srt = breaks.tolist()
        if not coherent:
            assert native > 4 * srt                                     # a shuffled list breaks its runs all the time
        # consecutive tets of the computed order are neighbours: the mean centroid step is a small fraction of the grid
        c = arr[o].mean(1)
        assert np.abs(np.diff(c, axis=0)).sum(1).mean() < 0.2
    # non-finite tets go to the end, the rest is still a permutation
    bad = shuffled.copy()
    bad[5, 2, 1] = np.nan
    bad[77] = np.inf
    o = hip_ops.tet_spatial_order(torch.from_numpy(bad).to(cuda)).cpu().numpy()
    assert np.array_equal(np.sort(o), np.arange(bad.shape[0])) and set(o[-2:]) == {5, 77}


@pytest.mark.parametrize("algo", [0, 3, 4, 5])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_ordered_traversal_bit_exact_adversarial(cuda, oracle, algo, seed):
    """Every special class (irregular tets, NaN queries, duplicates where the LOWEST index must win) through the ordered
    instances of the filter kernels, with the computed order and with a random permutation."""
    from deftet_amd import hip_ops
    tet, pts = cases.adversarial(seed)
    want = oracle.point_in_tet(tet, pts)
    t, p = torch.from_numpy(tet).to(cuda), torch.from_numpy(pts).to(cuda)
    T = tet.shape[1]
    perm = torch.from_numpy(np.random.default_rng(seed).permutation(T).astype(np.int32)).to(cuda)
    for order in (hip_ops.tet_spatial_order(t[0]), perm):
        got = hip_ops.point_in_tet(t, p, algo=algo, order=order)
        assert np.array_equal(got.cpu().numpy(), want)


def test_shuffled_tets_configs2_size_bit_exact_vs_brute(cuda):
    """BASELINE configs[2] size (res 70, 100k queries; two shapes), the tet list randomly shuffled: the caller's order, the
    computed traversal order and order="auto" all give the brute-force kernel's answer, the same weights and the same
    gradients."""
    from deftet_amd import grids, hip_ops
    tet, pts, _, _ = grids.make_case(70, 100_000, 2)
    rng = np.random.default_rng(11)
    tet = np.ascontiguousarray(tet[:, rng.permutation(tet.shape[1])])
    t, p = torch.from_numpy(tet).to(cuda), torch.from_numpy(pts).to(cuda)
    brute = hip_ops.point_in_tet(t, p, algo=hip_ops.PIT_BRUTE)
    order, breaks = hip_ops.tet_spatial_order(t[0], want_breaks=True)
    native, srt = breaks.tolist()
    assert native > 4 * srt
    hip_ops.clear_tet_order_cache()
    assert hip_ops.auto_tet_order(t, p) is not None                    # the coherence rule takes the computed order here
    (choice, fractions), = hip_ops.tet_order_decisions().values()
    assert choice == "sorted" and fractions[0] > 0.9 and fractions[1] < 0.05   # far steps: the caller's list, the computed order
    gw = torch.randn(2, 100_000, 4, device=cuda, generator=torch.Generator(device=cuda).manual_seed(0))
    for algo in (hip_ops.PIT_PAIR, hip_ops.PIT_WAVE, hip_ops.PIT_SLAB):
        ref = None
        for o in (None, order, "auto"):
            cond, w, hits = hip_ops.point_in_tet(t, p, want_bary=True, want_hits=True, algo=algo, order=o)
            assert torch.equal(cond, brute), (algo, o is None)
            g = hip_ops.point_in_tet_bwd(t, p, cond, gw, hits=hits)[0]
            if ref is None:
                ref = (w, g)
            assert torch.equal(w, ref[0]) and torch.equal(g, ref[1])   # recorded hits are summed in ascending query order
    hip_ops.clear_tet_order_cache()


def test_auto_order_is_decided_per_topology_from_coherence(cuda):
    """order="auto" (round 6): a deterministic rule on the numbering's coherence (far steps inside 64-tet groups), cached per
    TOPOLOGY — two meshes with the same number of tets and different numberings get different decisions, whichever comes first;
    the shipped QuarTet grid (76 % column changes, yet neighbours all the way) keeps its own numbering; results never change."""
    from deftet_amd import grids, hip_ops
    tet, pts, _, _ = grids.make_case(40, 20_000, 2)
    rng = np.random.default_rng(5)
    perm = rng.permutation(tet.shape[1])
    t_coh, t_shuf = torch.from_numpy(tet).to(cuda), torch.from_numpy(np.ascontiguousarray(tet[:, perm])).to(cuda)
    p = torch.from_numpy(pts).to(cuda)
    idx_coh = torch.arange(tet.shape[1] * 4, device=cuda).reshape(-1, 4)            # stand-ins for the two index lists
    idx_shuf = idx_coh[torch.from_numpy(perm).to(cuda)].contiguous()
    far_coh, far_shuf = hip_ops.tet_order_coherence(t_coh[0]).tolist(), hip_ops.tet_order_coherence(t_shuf[0]).tolist()
    assert far_coh[1] == far_shuf[1] == tet.shape[1] - (tet.shape[1] + 63) // 64
    assert far_coh[0] < 0.05 * far_coh[1] and far_shuf[0] > 0.9 * far_shuf[1]
    srt = hip_ops.tet_spatial_order(t_shuf[0])
    assert hip_ops.tet_order_coherence(t_shuf[0], srt).tolist()[0] < 0.05 * far_shuf[1]
    brute = {"coh": hip_ops.point_in_tet(t_coh, p, algo=hip_ops.PIT_BRUTE), "shuf": hip_ops.point_in_tet(t_shuf, p, algo=hip_ops.PIT_BRUTE)}
    for first in ("coh", "shuf"):
        hip_ops.clear_tet_order_cache()
        for name in ((first, "shuf" if first == "coh" else "coh") * 2):
            tt, topo = (t_coh, idx_coh) if name == "coh" else (t_shuf, idx_shuf)
            got = hip_ops.point_in_tet(tt, p, order="auto", topology=topo)
            assert torch.equal(got, brute[name]), (first, name)
        dec = hip_ops.tet_order_decisions()
        assert sorted(v[0] for v in dec.values()) == ["native", "sorted"], dec          # same T, same kernel: two entries, two decisions
        for (_, _, _, tk), (choice, fr) in dec.items():
            assert tk[0] == "tensor" and (choice == "sorted") == (fr[0] > 0.5)
    # a TetTopology-like object (serial) and a plain hashable key work as well
    hip_ops.clear_tet_order_cache()

    class Topo:
        serial = 12345
    assert hip_ops.auto_tet_order(t_shuf, p, topology=Topo()) is not None and hip_ops.auto_tet_order(t_coh, p, topology="kuhn40") is None
    # the shipped QuarTet grid: its numbering changes column at three steps out of four and is still the one to traverse
    g40 = np.load(os.path.join(os.path.dirname(__file__), "golden", "cube40_grid.npz"))
    q40 = torch.from_numpy(np.ascontiguousarray(g40["verts"].astype(np.float32)[g40["tets"]])[None]).to(cuda)
    _, breaks = hip_ops.tet_spatial_order(q40[0], want_breaks=True)
    far = hip_ops.tet_order_coherence(q40[0]).tolist()
    assert breaks.tolist()[0] > 0.5 * far[1] and far[0] < 0.10 * far[1], (breaks.tolist(), far)
    assert hip_ops.auto_tet_order(q40, p[:1], topology="cube40") is None
    hip_ops.clear_tet_order_cache()


def test_auto_order_without_a_topology_notices_another_mesh(cuda):
    """Callers with positions only (check_condition_f_base's reference signature): the decision is keyed by the sizes and WATCHED —
    a second mesh with the same number of tets and another numbering is noticed within a few watch periods and decided again;
    every call along the way returns the brute-force answer."""
    from deftet_amd import grids, hip_ops
    tet, pts, _, _ = grids.make_case(40, 20_000, 1)
    perm = np.random.default_rng(6).permutation(tet.shape[1])
    t_coh, t_shuf = torch.from_numpy(tet).to(cuda), torch.from_numpy(np.ascontiguousarray(tet[:, perm])).to(cuda)
    p = torch.from_numpy(pts).to(cuda)
    brute = {id(t_coh): hip_ops.point_in_tet(t_coh, p, algo=hip_ops.PIT_BRUTE), id(t_shuf): hip_ops.point_in_tet(t_shuf, p, algo=hip_ops.PIT_BRUTE)}
    hip_ops.clear_tet_order_cache()

    def decision():
        (choice, _), = hip_ops.tet_order_decisions().values()
        return choice

    for tt, want in ((t_coh, "native"), (t_shuf, "sorted"), (t_coh, "native")):
        for i in range(4 * hip_ops._WATCH_EVERY + 2):
            got = hip_ops.point_in_tet(tt, p, order="auto")
            if i % 50 == 0:
                assert torch.equal(got, brute[id(tt)])
                torch.cuda.synchronize()
        assert decision() == want, (want, hip_ops.tet_order_decisions())
    hip_ops.clear_tet_order_cache()


def test_order_argument_must_be_a_permutation(cuda):
    """A caller's `order` is validated once per tensor: duplicates would skip tets, entries outside [0, T) index the
    tet array and the hit records out of bounds."""
    from deftet_amd import grids, hip_ops
    tet, pts, _, _ = grids.make_case(8, 500, 1)
    t, p = torch.from_numpy(tet).to(cuda), torch.from_numpy(pts).to(cuda)
    T = tet.shape[1]
    good = torch.from_numpy(np.random.default_rng(0).permutation(T).astype(np.int32)).to(cuda)
    assert torch.equal(hip_ops.point_in_tet(t, p, order=good), hip_ops.point_in_tet(t, p))
    for bad in (good.clone().index_fill_(0, torch.tensor([3], device=cuda), int(good[4])),      # a duplicate
                good.clone().index_fill_(0, torch.tensor([0], device=cuda), T),                 # past the end
                good.clone().index_fill_(0, torch.tensor([T - 1], device=cuda), -1)):           # negative
        with pytest.raises(RuntimeError, match="permutation"):
            hip_ops.point_in_tet(t, p, order=bad)
    good[0], good[1] = int(good[1]), int(good[0])                       # modified in place: checked again, still a permutation
    assert torch.equal(hip_ops.point_in_tet(t, p, order=good), hip_ops.point_in_tet(t, p))


def test_wave_kernel_clamped_footprints_and_degenerate_axis(cuda, oracle):
    """Directed at k_tet_scan_wave's radius bound: most tets straddle or lie outside the query bounding
    box (their cell footprints are clamped), and in a second case all queries share one x coordinate (inv == 0 on that axis).
    Compared bit-exact against the exact binned kernel, the brute kernel and the oracle."""
    from deftet_amd import grids, hip_ops
    tet, _, _, _ = grids.make_case(16, 10, 2)
    rng = np.random.default_rng(5)
    for case in range(3):
        if case == 0:       # queries in a small box in the middle: most tets are outside, many straddle its faces
            pts = (rng.random((2, 6000, 3)).astype(np.float32) - 0.5) * 0.22 + 0.03
        elif case == 1:     # coplanar queries: the x extent of their box is zero
            pts = (rng.random((2, 6000, 3)).astype(np.float32) - 0.5) * 0.9
            pts[..., 0] = 0.0625
        else:               # collinear queries (two degenerate axes), crossing the whole grid and leaving it
            pts = np.zeros((2, 3000, 3), np.float32)
            pts[..., 2] = np.linspace(-0.7, 0.7, 3000, dtype=np.float32)
            pts[..., 0] = 0.03125
            pts[..., 1] = -0.125
        want = oracle.point_in_tet(tet, pts)
        t, p = torch.from_numpy(tet).to(cuda), torch.from_numpy(pts).to(cuda)
        for algo in (1, 2, 3, 4, 5):
            assert np.array_equal(hip_ops.point_in_tet(t, p, algo=algo).cpu().numpy(), want), (case, algo)
        assert (want >= 0).any()


# ---------------------------------------------------------------------------------------------------------------------
# query box (round 5): a box handed in replaces the measuring launch; it is a hint and must never change a result
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("algo", [0, 2, 3, 4, 5])
def test_query_box_hint_never_changes_the_result(cuda, oracle, algo):
    from deftet_amd import hip_ops
    tet, pts = cases.jittered(12, 5000, 2)
    pts = pts.copy()
    pts[0, 7] = np.nan
    pts[1, 11, 2] = np.inf
    pts[1, 12] = 3e7
    want = oracle.point_in_tet(tet, pts)
    t, p = torch.from_numpy(tet).to(cuda), torch.from_numpy(pts).to(cuda)
    gw = torch.randn(2, 5000, 4, device=cuda, generator=torch.Generator(device=cuda).manual_seed(1))
    ref_c, ref_w, ref_h = hip_ops.point_in_tet(t, p, want_bary=True, want_hits=True, algo=algo)
    assert np.array_equal(ref_c.cpu().numpy(), want)
    ref_g = hip_ops.point_in_tet_bwd(t, p, ref_c, gw, hits=ref_h)[0]
    fin = torch.from_numpy(np.nan_to_num(pts, nan=0.0, posinf=0.0, neginf=0.0).clip(-1, 1)).to(cuda)
    exact = torch.cat([fin.amin(1), fin.amax(1)], 1).contiguous()
    boxes = {
        "exact": exact,
        "half": (exact * 0.5).contiguous(),                                       # most queries outside: the side path answers them
        "tiny": (exact * 1e-3).contiguous(),
        "elsewhere": (exact + 5.0).contiguous(),                                  # no query inside
        "nan": torch.full((2, 6), float("nan"), device=cuda),
        "inverted": torch.cat([exact[:, 3:], exact[:, :3]], 1).contiguous(),
        "huge": torch.tensor([[-1e30] * 3 + [1e30] * 3] * 2, device=cuda),
        "flat": torch.cat([exact[:, :3], exact[:, :2], exact[:, 2:3]], 1).contiguous(),   # zero extent along z
    }
    for name, box in boxes.items():
        c, w, h = hip_ops.point_in_tet(t, p, want_bary=True, want_hits=True, algo=algo, query_box=box)
        assert np.array_equal(c.cpu().numpy(), want), name
        assert torch.equal(w.view(torch.int32), ref_w.view(torch.int32)), name     # (bit patterns: the NaN query hits tet 0 with NaN weights)
        g = hip_ops.point_in_tet_bwd(t, p, c, gw, hits=h)[0]
        fin_g = torch.isfinite(ref_g)
        assert torch.equal(fin_g, torch.isfinite(g)), name
        assert (g - ref_g)[fin_g].abs().max().item() <= 2e-6 * ref_g[fin_g].abs().max().item(), name   # (side-path hits are summed by the list path)


def test_query_box_misses_are_counted_exactly(cuda):
    """query_box_misses: the number of REGULAR queries per shape outside the hinted box (enlarged by 1/32 per side, as the grid
    spans it) — into device memory or into pinned host memory the kernel writes directly (what the tracker polls); NaN / Inf /
    huge queries are not misses of the box."""
    from deftet_amd import hip_ops
    tet, pts = cases.jittered(12, 20000, 3)
    pts = pts.copy()
    pts[0, :50] = np.nan
    pts[1, 100:130, 1] = np.inf
    pts[2, 7] = 3e7
    t, p = torch.from_numpy(tet).to(cuda), torch.from_numpy(pts).to(cuda)
    ref = hip_ops.point_in_tet(t, p)
    big = 1048576.0
    regular = (p.abs() <= big).all(-1)
    fin = torch.where(regular[..., None], p, torch.zeros_like(p))
    exact = torch.cat([fin.amin(1), fin.amax(1)], 1).contiguous()
    for name, box in (("exact", exact), ("half", (exact * 0.5).contiguous()), ("shifted", (exact + 0.3).contiguous()),
                      ("elsewhere", (exact + 5.0).contiguous())):
        lo, hi = box[:, :3], box[:, 3:]
        e = (hi - lo) * (1.0 / 32.0)
        glo, ghi = torch.clamp(lo - e, min=-big), torch.clamp(hi + e, max=big)
        inside = ((p >= glo[:, None]) & (p <= ghi[:, None])).all(-1)
        want = (regular & ~inside).sum(1).to(torch.int32)
        on_dev = torch.full((3,), -1, device=cuda, dtype=torch.int32)
        on_host = torch.full((3,), -1, dtype=torch.int32).pin_memory()
        for algo in (0, 3):
            got = hip_ops.point_in_tet(t, p, algo=algo, query_box=box, query_box_misses=on_dev)
            assert torch.equal(got, ref), name
            got = hip_ops.point_in_tet(t, p, algo=algo, query_box=box, query_box_misses=on_host)
            assert torch.equal(got, ref), name
        torch.cuda.synchronize()
        assert torch.equal(on_dev.cpu(), want.cpu()), (name, on_dev.tolist(), want.tolist())
        assert torch.equal(on_host, want.cpu()), (name, on_host.tolist(), want.tolist())
        if name == "exact":
            assert int(want.sum()) == 0
        else:
            assert int(want.min()) > 0


def test_query_box_tracking_backs_off_when_the_queries_stop_fitting(cuda):
    """query_box="track" with a caller that alternates between two query distributions under the same (B, Q): every call exact;
    the tracker sees the misses in its pinned mailbox and goes back to measuring, for longer each time (the side path costs a
    brute-force scan per query outside the box, so tracking must not go on blindly)."""
    from deftet_amd import grids, hip_ops
    tet, _, _, _ = grids.make_case(16, 10, 2)
    t = torch.from_numpy(tet).to(cuda)
    hip_ops.clear_query_box_cache()
    rng = np.random.default_rng(3)
    near = torch.from_numpy((0.5 * (rng.random((2, 3000, 3)) - 0.5)).astype(np.float32)).to(cuda)
    wide = torch.from_numpy((1.05 * (rng.random((2, 3000, 3)) - 0.5)).astype(np.float32)).to(cuda)
    refs = [hip_ops.point_in_tet(t, q, algo=hip_ops.PIT_BRUTE) for q in (near, wide)]
    key = hip_ops.query_box_key(cuda, 2, 3000)
    for i in range(40):
        got = hip_ops.point_in_tet(t, (near, wide)[i % 2], query_box="track")
        assert torch.equal(got, refs[i % 2]), i
        torch.cuda.synchronize()                                   # (the mailbox is polled without one: here every call sees the last)
    st = hip_ops.query_box_trackers()[key]
    assert st["backoffs"] >= 2 and st["measured"] >= 30 and st["tracked"] <= 8, st
    # one distribution from now on: tracking resumes and stays
    before = st["tracked"]
    for i in range(st["hold"] + 12):
        assert torch.equal(hip_ops.point_in_tet(t, wide, query_box="track"), refs[1])
        torch.cuda.synchronize()
    st = hip_ops.query_box_trackers()[key]
    assert st["tracked"] >= before + 10 and st["hold"] == 0, st
    hip_ops.clear_query_box_cache()


def test_query_box_tracking_over_rotating_query_sets(cuda):
    """query_box="track" as the autograd ops use it: three different query sets in turns (different boxes), every call equal to the
    brute-force kernel; then a set from a shifted, larger box (the first call after the shift is served by the side path)."""
    from deftet_amd import grids, hip_ops
    tet, _, _, _ = grids.make_case(16, 10, 2)
    t = torch.from_numpy(tet).to(cuda)
    hip_ops.clear_query_box_cache()
    rng = np.random.default_rng(0)
    sets = [torch.from_numpy((1.05 * (rng.random((2, 4000, 3)) - 0.5)).astype(np.float32)).to(cuda) for _ in range(3)]
    for i in range(7):
        p = sets[i % 3]
        got = hip_ops.point_in_tet(t, p, query_box="track")
        assert torch.equal(got, hip_ops.point_in_tet(t, p, algo=hip_ops.PIT_BRUTE)), i
    shifted = (sets[0] * 1.6 + 0.1).contiguous()
    for i in range(3):
        got = hip_ops.point_in_tet(t, shifted, query_box="track")
        assert torch.equal(got, hip_ops.point_in_tet(t, shifted, algo=hip_ops.PIT_BRUTE)), i
    # the two-call form takes the box on the query side
    pq = hip_ops.prepare_queries(sets[1], t.shape[1], query_box="track")
    got = hip_ops.point_in_tet(t, sets[1], prepared=pq)
    assert torch.equal(got, hip_ops.point_in_tet(t, sets[1], algo=hip_ops.PIT_BRUTE))
    hip_ops.clear_query_box_cache()


def test_step_captured_in_a_hipgraph_replays_on_new_inputs(cuda):
    """fwd (index, weights, occupancy, hit records) + bwd + loss row dots captured in ONE hipGraph with the hints the autograd ops pass
    (order="auto", query_box="track") — as the very first call of its (B, Q), so the tracker must not create state inside the
    capture — then replayed on other inputs copied into the captured tensors: every replay equals the eager operator."""
    from deftet_amd import grids, hip_ops
    hip_ops.clear_query_box_cache()
    B, Q = 2, 2500                                                   # a (B, Q) no other test tracks; <= 2 queries per tet: the backward reads the records
    sets = []
    for s in range(3):
        tet, pts, _, _ = grids.make_case(12, Q, B, 0.1 + 0.05 * s)
        g = torch.Generator(device=cuda).manual_seed(50 + s)
        t = torch.from_numpy(tet).to(cuda)
        sets.append(dict(tet=t, pts=torch.from_numpy(pts).to(cuda) * (1.0 + 0.1 * s), pred=torch.rand(B, t.shape[1], device=cuda, generator=g),
                         gw=torch.randn(B, Q, 4, device=cuda, generator=g), go=torch.randn(B, Q, device=cuda, generator=g)))
    st = {k: v.clone() for k, v in sets[0].items()}                 # the tensors the graph reads

    def step(d, **hints):
        cond, w, occ, hits = hip_ops.point_in_tet(d["tet"], d["pts"], want_bary=True, pred_bxt=d["pred"], want_hits=True, **hints)
        loss = hip_ops.rowdot(w, d["gw"], occ, d["go"])
        gt, _, gp = hip_ops.point_in_tet_bwd(d["tet"], d["pts"], cond, d["gw"], grad_occ=d["go"], hits=hits)
        return cond, w, occ, loss, gt, gp

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step(st)                                                     # workspaces reach their size outside the capture (no hints: no tracker yet)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step(st, order="auto", query_box="track")
    assert not any(k[0] == (cuda.index or 0) and k[2:] == (B, Q) for k in hip_ops.query_box_trackers())                   # the capture measured its box: nothing was allocated for tracking
    for i in (1, 2, 0, 2):
        for k in st:
            st[k].copy_(sets[i][k])
        graph.replay()
        torch.cuda.synchronize()
        want = step(sets[i])
        assert torch.equal(out[0], hip_ops.point_in_tet(sets[i]["tet"], sets[i]["pts"], algo=hip_ops.PIT_BRUTE)), i
        for name, a, b in zip(("cond", "w", "occ", "loss", "grad_tet", "grad_pred"), out, want):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (i, name)
    # ... and with a tracker that already exists a captured step still measures its own box and leaves the tracker alone (round 6:
    # a tracked triple baked into a graph would never alternate its buffers nor ever fall back to measuring)
    for i in range(3):
        step(sets[i], order="auto", query_box="track")
    before = hip_ops.query_box_trackers()
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2):
        out2 = step(st, order="auto", query_box="track")
    assert hip_ops.query_box_trackers() == before
    for i in (2, 1):
        for k in st:
            st[k].copy_(sets[i][k])
        graph2.replay()
        torch.cuda.synchronize()
        want = step(sets[i])
        for name, a, b in zip(("cond", "w", "occ", "loss", "grad_tet", "grad_pred"), out2, want):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (i, name)
    del graph, graph2
    hip_ops.clear_query_box_cache()


def test_backward_is_bit_reproducible_with_overflowed_records(cuda):
    """Half of the queries in an eighth of the volume (8 to 9 per tet there, under 2 per tet overall: the backward reads the records):
    many tets accept more queries than their hit record and its spill record hold, so their hits come from the forward's list of
    unrecorded hits, whose ORDER changes from run to run (workgroups append to it with an atomic).  The backward adds them in
    ascending query order all the same: every run gives the same bits (rounds 2-4 did not: the lanes of a butterfly held the hits
    in list order)."""
    from deftet_amd import grids, hip_ops
    for res, Q in ((12, 2500), (20, 11000)):
        tet, pts, _, _ = grids.make_case(res, Q, 2, 0.1)
        pts = pts.copy()
        pts[:, : Q // 2] *= 0.5
        t, p = torch.from_numpy(tet).to(cuda), torch.from_numpy(pts).to(cuda)
        g = torch.Generator(device=cuda).manual_seed(5)
        gw, go = torch.randn(2, Q, 4, device=cuda, generator=g), torch.randn(2, Q, device=cuda, generator=g)
        pred = torch.rand(2, t.shape[1], device=cuda, generator=g)
        ref = None
        for it in range(12):
            cond, w, occ, hits = hip_ops.point_in_tet(t, p, want_bary=True, pred_bxt=pred, want_hits=True)
            gt, gq, gp = hip_ops.point_in_tet_bwd(t, p, cond, gw, want_grad_pts=True, grad_occ=go, hits=hits)
            if it == 0:
                assert hip_ops.point_in_tet_stats(2, t.shape[1], Q, 0, cuda)[:, 6].min() > 0      # overflowed tets in every shape
                ref = (gt.clone(), gq.clone(), gp.clone())
                lst = hip_ops.point_in_tet_bwd(t, p, cond, gw, want_grad_pts=True, grad_occ=go)   # the list backward: same sums up to rounding
                assert (lst[0] - gt).abs().max().item() <= 2e-6 * gt.abs().max().item()
            else:
                for name, a, b in zip(("grad_tet", "grad_pts", "grad_pred"), (gt, gq, gp), ref):
                    assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (res, it, name)


@pytest.mark.parametrize("algo", [0, 3, 4])
def test_shapes_of_a_batch_do_not_see_each_other(cuda, algo):
    """A batch is B independent problems (per-shape grids, sorts, records, lists): every shape gives the same bits alone and inside
    a batch of different shapes — index, weights, occupancy, all gradients — also when one of the others is full of NaN queries,
    has a query set of another scale, or tets that overflow their records."""
    from deftet_amd import grids, hip_ops
    Q = 2500                                                         # (<= 2 queries per tet on average: the backward reads the records)
    tet, pts, _, _ = grids.make_case(12, Q, 4, 0.15)
    pts = pts.copy()
    pts[1, :500] = np.nan
    pts[2] *= 3.0                                                    # another box: most of these queries miss
    pts[3] = pts[3] * 0.6                                            # dense: 9 queries per tet in the middle, those records overflow
    t, p = torch.from_numpy(tet).to(cuda), torch.from_numpy(pts).to(cuda)
    g = torch.Generator(device=cuda).manual_seed(9)
    gw, go = torch.randn(4, Q, 4, device=cuda, generator=g), torch.randn(4, Q, device=cuda, generator=g)
    pred = torch.rand(4, t.shape[1], device=cuda, generator=g)

    def run(sl):
        cond, w, occ, hits = hip_ops.point_in_tet(t[sl], p[sl], want_bary=True, pred_bxt=pred[sl], want_hits=True, algo=algo)
        gt, gq, gp = hip_ops.point_in_tet_bwd(t[sl], p[sl], cond, gw[sl], want_grad_pts=True, grad_occ=go[sl], hits=hits)
        return cond, w, occ, gt, gq, gp

    whole = run(slice(0, 4))
    assert hip_ops.point_in_tet_stats(4, t.shape[1], Q, algo, cuda)[3, 6] > 0          # shape 3 has overflowed records
    for b in range(4):
        alone = run(slice(b, b + 1))
        for name, x, y in zip(("cond", "w", "occ", "grad_tet", "grad_pts", "grad_pred"), whole, alone):
            hitq = (whole[0][b, :, 0] >= 0)
            xa, ya = x[b:b + 1], y
            if name == "grad_pts":                                   # defined for the hits only (a miss has no tet to differentiate through)
                xa, ya = xa[0][hitq], ya[0][hitq]
            assert torch.equal(xa.contiguous().view(torch.int32), ya.contiguous().view(torch.int32)), (b, name)


def test_autograd_ops_in_the_dense_case(cuda, oracle):
    """13 queries per tet: the autograd operators do not ask the forward for hit records (hip_ops.bwd_uses_records) and the backward
    takes the per-tet lists — same index / weights / occupancy as ever, gradients equal to the fp64 autograd of the reference's
    barycentric formula."""
    from deftet_amd import hip_ops
    from deftet_amd.layers.DefTet.check_condition_tetrahedron_base.utils import point_in_tet_occ
    tet, pts = cases.jittered(8, 5000, 2)
    assert not hip_ops.bwd_uses_records(tet.shape[1], pts.shape[1]) and hip_ops.bwd_uses_records(257250, 100000)
    t = torch.from_numpy(tet).to(cuda).requires_grad_(True)
    p = torch.from_numpy(pts).to(cuda)
    g = torch.Generator(device=cuda).manual_seed(2)
    pred = torch.rand(2, tet.shape[1], device=cuda, generator=g).requires_grad_(True)
    gw, go = torch.randn(2, 5000, 4, device=cuda, generator=g), torch.randn(2, 5000, device=cuda, generator=g)
    cond, w, occ = point_in_tet_occ(t, p, pred)
    assert np.array_equal(cond.detach().cpu().numpy(), oracle.point_in_tet(tet, pts))
    ((w * gw).sum() + (occ * go).sum()).backward()
    # fp64 reference: the reference's bary_centric_tet on the winning tets, paste_occ with misses aliasing tet 0
    t64 = torch.from_numpy(tet).double().to(cuda).requires_grad_(True)
    p64 = pred.detach().double().requires_grad_(True)
    idx = cond.detach()[..., 0].long()
    hit = idx >= 0
    sel = torch.gather(t64, 1, idx.clamp_min(0)[:, :, None, None].expand(-1, -1, 4, 3))
    a, b, c, d = sel.unbind(2)
    q = p.double()
    vol = lambda x, y, z: (x * torch.cross(y, z, dim=-1)).sum(-1)
    v6 = 1.0 / vol(b - a, c - a, d - a)
    w64 = torch.stack([vol(q - b, d - b, c - b) * v6, vol(q - a, c - a, d - a) * v6, vol(q - a, d - a, b - a) * v6, vol(q - a, b - a, c - a) * v6], -1)
    w64 = torch.where(hit[..., None], w64, torch.zeros_like(w64))
    occ64 = torch.gather(p64, 1, idx.clamp_min(0))
    ((w64 * gw.double()).sum() + (occ64 * go.double()).sum()).backward()
    check_close("dense w", w.detach().double(), w64.detach(), 2e-6)
    check_close("dense grad_tet", t.grad.double(), t64.grad, 4e-6)
    check_close("dense grad_pred", pred.grad.double(), p64.grad, 2e-6)


@pytest.mark.parametrize("algo", [0, 2, 3, 4, 5])
def test_wide_tets_are_tested_by_the_query_lanes(cuda, algo):
    """All queries inside a box the size of a tet or smaller (or at one point): the grid spans that box, so every tet that touches it
    covers a large part of the cells and would walk thousands of candidates in one lane.  Such tets are handed to k_finalize like
    the irregular ones (counters word 0) — same index, weights and gradients as the brute-force kernel / the list backward."""
    from deftet_amd import grids, hip_ops
    Q = 20000
    tet, pts, _, _ = grids.make_case(16, Q, 3, 0.1)
    pts = pts.copy()
    pts[0] = pts[0] * 1e-3 + 0.1                                      # a ball far smaller than a tet
    pts[1] = pts[1] * 0.08 - 0.2                                      # about one tet wide
    pts[2] = 0.25                                                    # one point: every axis of the grid is degenerate
    t, p = torch.from_numpy(tet).to(cuda), torch.from_numpy(pts).to(cuda)
    g = torch.Generator(device=cuda).manual_seed(4)
    gw = torch.randn(3, Q, 4, device=cuda, generator=g)
    ref = hip_ops.point_in_tet(t, p, algo=hip_ops.PIT_BRUTE)
    cond, w, hits = hip_ops.point_in_tet(t, p, want_bary=True, want_hits=True, algo=algo)
    assert torch.equal(cond, ref)
    assert (ref >= 0).float().mean().item() > 0.9
    stats = hip_ops.point_in_tet_stats(3, t.shape[1], Q, algo, cuda)
    assert (stats[:, 0] > 0).all(), stats[:, 0]                       # tets listed for k_finalize: the wide ones (the mesh has no irregular tet)
    a = hip_ops.point_in_tet_bwd(t, p, cond, gw, hits=hits)[0]
    b = hip_ops.point_in_tet_bwd(t, p, cond, gw)[0]
    check_close("wide tets: hit-record backward vs list backward", a, b, 2e-5)
    # the uniform case of the same size lists none
    u = torch.from_numpy(grids.make_case(16, Q, 3, 0.1)[1]).to(cuda)
    hip_ops.point_in_tet(t, u, algo=algo)
    assert (hip_ops.point_in_tet_stats(3, t.shape[1], Q, algo, cuda)[:, 0] == 0).all()
