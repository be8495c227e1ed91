"""GPU tests of the tet-face rasterizer (A12).  PARITY UNPINNED w.r.t. Kaolin (not in the
reference tree); these tests pin the HIP implementation to this repo's statement of the
contract (oracle/deftet_oracle_render.c), to fp64 autograd for the backward, and to
size-independent properties at the BASELINE configs[4] size."""
import numpy as np
import pytest
import torch

from deftet_amd import grids

pytestmark = pytest.mark.gpu


def projected_grid(res, **kw):
    """Unique faces (incl. boundary) of a res=R Kuhn grid (from the ORACLE's face table), projected by
    grids.project_faces.  Returns face_z [1,F,3], face_xy [1,F,3,2], feat [1,F,3,4]."""
    from oracle import oracle as O
    verts, tets = grids.kuhn_grid(res)
    f3, _, _, _, _ = O.tet_to_face(tets, verts.shape[0], with_boundary=True)
    return grids.project_faces(verts, f3, **kw)


pixel_grid = grids.pixel_grid


NEAREST, FIRST = 0, 1         # saturation policies (include/deftet_hip.h)


def run(pix, rngs, fz, fxy, ff, knum, dev, eps=1e-8, policy=NEAREST):
    from deftet_amd.render import deftet_sparse_render
    t = [torch.from_numpy(x).to(dev) for x in (pix, rngs, fz, fxy, ff)]
    feat, face = deftet_sparse_render(*t, knum=knum, eps=eps, policy=policy)
    torch.cuda.synchronize()
    return feat, face


@pytest.mark.parametrize("policy", [NEAREST, FIRST])
@pytest.mark.parametrize("knum", [4, 64])
def test_forward_matches_oracle_projected_grid(cuda, oracle, knum, policy):
    fz, fxy, ff = projected_grid(8)
    pix, rngs = pixel_grid(40)
    pix = pix * 0.7
    wf, wface, ww = oracle.sparse_render_fwd(pix, rngs, fz, fxy, ff, knum=knum, policy=policy)
    feat, face = run(pix, rngs, fz, fxy, ff, knum, cuda, policy=policy)
    assert np.array_equal(face.cpu().numpy(), wface)
    assert np.array_equal(feat.cpu().numpy(), wf)
    nh = (wface >= 0).sum(-1)
    assert nh.max() == knum if knum == 4 else nh.max() > 20      # knum=4 overflows: the saturation policy decides


def test_saturation_policies_differ_only_where_pixels_saturate(cuda, oracle):
    """NEAREST keeps the knum nearest covering faces, FIRST the first knum in face order: identical rows wherever a pixel
    has at most knum covering faces, and on saturated pixels NEAREST's record is the head of the unbounded (knum = F)
    record while FIRST's is the sorted head of the face-order prefix."""
    fz, fxy, ff = projected_grid(8)
    pix, rngs = pixel_grid(40)
    pix = pix * 0.7
    k = 12
    full_feat, full_face = run(pix, rngs, fz, fxy, ff, 120, cuda)             # no pixel has 120 covering faces here
    n_cover = (full_face >= 0).sum(-1)
    assert n_cover.max().item() < 120 and (n_cover > k).any() and (n_cover <= k).any()
    near_feat, near_face = run(pix, rngs, fz, fxy, ff, k, cuda, policy=NEAREST)
    first_feat, first_face = run(pix, rngs, fz, fxy, ff, k, cuda, policy=FIRST)
    assert torch.equal(near_face, full_face[..., :k]) and torch.equal(near_feat, full_feat[..., :k, :])
    unsat = n_cover <= k
    assert torch.equal(first_face[unsat], near_face[unsat]) and torch.equal(first_feat[unsat], near_feat[unsat])
    sat = ~unsat
    # FIRST: the k smallest face indices among the covering faces
    cover = torch.where(full_face >= 0, full_face, torch.full_like(full_face, 1 << 40))
    smallest = torch.sort(cover, dim=-1).values[..., :k]
    assert torch.equal(torch.sort(first_face[sat], dim=-1).values, smallest[sat])
    assert (first_face[sat] != near_face[sat]).any()


def test_forward_matches_oracle_adversarial(cuda, oracle):
    rng = np.random.default_rng(3)
    F = 400
    fxy = rng.uniform(-1, 1, (1, F, 3, 2)).astype(np.float32)
    fxy[0, :150] = fxy[0, :150] * 0.1 + rng.uniform(-0.9, 0.9, (150, 1, 2)).astype(np.float32)     # small faces (tiles)
    fxy[0, 150:160, 2] = fxy[0, 150:160, 0]                                                          # zero area
    fxy[0, 160:165, 2] = (fxy[0, 160:165, 0] + fxy[0, 160:165, 1]) / 2                               # collinear
    fxy[0, 165, 0, 0] = np.nan
    fxy[0, 166, 1] = np.inf
    fxy[0, 167] *= 1e7
    fxy[0, 168] = fxy[0, 3]                                                                          # duplicate face
    fxy[0, 169] = fxy[0, 5][::-1]                                                                    # reversed winding
    fz = rng.uniform(-5, -1, (1, F, 3)).astype(np.float32)
    fz[0, 170:175] = 5.0                                                                             # outside the depth range
    ff = rng.random((1, F, 3, 5)).astype(np.float32)
    P = 1500
    pix = rng.uniform(-1.1, 1.1, (1, P, 2)).astype(np.float32)
    pix[0, :100] = fxy[0, rng.integers(0, 150, 100), rng.integers(0, 3, 100)]                        # on vertices
    pix[0, 100] = np.nan
    pix[0, 101, 0] = np.inf
    pix[0, 102] = 3e6
    rngs = np.tile(np.array([-1000.0, 0.0], np.float32), (1, P, 1))
    rngs[0, 200:300] = [-3.0, -2.0]                                                                  # narrow depth window
    for knum in (8, 300):
        for policy in (NEAREST, FIRST):
            wf, wface, ww = oracle.sparse_render_fwd(pix, rngs, fz, fxy, ff, knum=knum, policy=policy)
            feat, face = run(pix, rngs, fz, fxy, ff, knum, cuda, policy=policy)
            assert np.array_equal(face.cpu().numpy(), wface)
            assert np.array_equal(feat.cpu().numpy(), wf, equal_nan=True)


def test_alpha_composite_on_gpu_equals_reference_peel2mask(cuda):
    """The compositing step on the GPU against the reference-generated fixture (tests/golden/render_glue.npz: peel2mask of
    5_rendereq/deftetrneder.py:31-64 run on these layers), value and gradient w.r.t. the layers (fp64 autograd of the same
    expression as the gradient reference)."""
    import os
    from deftet_amd.render import alpha_composite
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "render_glue.npz"))
    ims = torch.from_numpy(g["peel_ims"]).to(cuda).requires_grad_(True)
    dep = torch.from_numpy(g["peel_depth"]).to(cuda)
    c, v, d = alpha_composite(ims, dep)
    np.testing.assert_allclose(c.detach().cpu().numpy(), g["peel_color"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(v.detach().cpu().numpy(), g["peel_vis"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(d.detach().cpu().numpy(), g["peel_dep"], rtol=1e-6, atol=2e-6)
    w = torch.rand_like(c)
    (c * w).sum().backward()
    i64 = ims.detach().double().requires_grad_(True)
    c64, _, _ = alpha_composite(i64, dep.double())
    (c64 * w.double()).sum().backward()
    # (batch 1 only: batch 0 holds opacities AT the clamp, where 1 - (1 - 1e-10) is 0 in fp32 and 1e-10 in fp64)
    assert (ims.grad[1].double() - i64.grad[1]).abs().max().item() <= 1e-5 * i64.grad[1].abs().max().item()


def test_backward_matches_fp64_autograd(cuda, oracle):
    from deftet_amd.render import deftet_sparse_render, alpha_composite
    fz, fxy, ff = projected_grid(6)
    pix, rngs = pixel_grid(24)
    pix = pix * 0.6
    tp, tr, tz = (torch.from_numpy(x).to(cuda) for x in (pix, rngs, fz))
    txy = torch.from_numpy(fxy).to(cuda).requires_grad_(True)
    tff = torch.from_numpy(ff).to(cuda).requires_grad_(True)
    feat, face = deftet_sparse_render(tp, tr, tz, txy, tff, knum=48)
    color, vis, _ = alpha_composite(feat)                  # front-to-back compositing on top (deftetrneder.py:102-113)
    g = torch.Generator(device=cuda).manual_seed(0)
    loss = (color * torch.rand(color.shape, device=cuda, generator=g)).sum() + (vis ** 2).sum()
    loss.backward()
    # fp64 reference: same face indices, differentiable interpolation, same compositing
    xy64 = torch.from_numpy(fxy).double().requires_grad_(True)
    ff64 = torch.from_numpy(ff).double().requires_grad_(True)
    feat64 = oracle.sparse_render_torch(torch.from_numpy(pix).double(), xy64, ff64, face.cpu())
    assert torch.allclose(feat64.float(), feat.detach().cpu(), rtol=1e-4, atol=1e-5)
    c64, v64, _ = alpha_composite(feat64)
    g = torch.Generator(device=cuda).manual_seed(0)
    wts = torch.rand(color.shape, device=cuda, generator=g).cpu().double()
    ((c64 * wts).sum() + (v64 ** 2).sum()).backward()
    from tests.tol import check_close
    for nm, got, want in (("xy", txy.grad, xy64.grad), ("feat", tff.grad, ff64.grad)):
        assert want.abs().max().item() > 0
        check_close("A12 grad_%s through compositing, res6 24x24 k48 vs fp64 autograd" % nm, got, want, 6e-7, elem_rel=1.5e-4)
    # faces that no pixel hit get exactly zero gradient
    hit = torch.zeros(fxy.shape[1], dtype=torch.bool)
    hit[face.cpu()[face.cpu() >= 0]] = True
    assert (txy.grad.cpu()[0][~hit] == 0).all() and (tff.grad.cpu()[0][~hit] == 0).all()


@pytest.mark.parametrize("D,B", [(3, 2), (6, 1), (9, 2)])
def test_backward_other_feature_widths_and_batches(cuda, oracle, D, B):
    """run-time feature width (the D != 4 kernel, several 4-channel chunks) and B > 1 against fp64 autograd"""
    from deftet_amd.render import deftet_sparse_render
    rng = np.random.default_rng(D * 10 + B)
    fz, fxy, _ = projected_grid(6)
    pix, rngs = pixel_grid(40)
    rep = lambda a: np.concatenate([a * (1 + 0.03 * b) for b in range(B)], 0)
    fz, fxy, pix = rep(fz), rep(fxy), rep(pix * 0.6)
    rngs = np.concatenate([rngs] * B, 0)
    ff = rng.random((B, fxy.shape[1], 3, D)).astype(np.float32)
    tp, tr, tz = (torch.from_numpy(x).to(cuda) for x in (pix, rngs, fz))
    txy = torch.from_numpy(fxy).to(cuda).requires_grad_(True)
    tff = torch.from_numpy(ff).to(cuda).requires_grad_(True)
    feat, face = deftet_sparse_render(tp, tr, tz, txy, tff, knum=40)
    go = torch.rand(feat.shape, device=cuda, generator=torch.Generator(device=cuda).manual_seed(3))
    gxy, gff = torch.autograd.grad(feat, (txy, tff), go)
    xy64 = txy.detach().double().requires_grad_(True)
    ff64 = tff.detach().double().requires_grad_(True)
    feat64 = oracle.sparse_render_torch(tp.double(), xy64, ff64, face)
    wxy, wff = torch.autograd.grad(feat64, (xy64, ff64), go.double())
    from tests.tol import check_close
    for nm, got, want in (("xy", gxy, wxy), ("feat", gff, wff)):
        check_close("A12 grad_%s, D=%d B=%d vs fp64 autograd" % (nm, D, B), got, want, 6e-7, elem_rel=3e-5)
    assert (face >= 0).sum().item() > 1000 * B


@pytest.mark.parametrize("npix,knum", [(23, 3), (37, 5), (25, 1), (31, 64)])
def test_backward_four_hits_per_lane_kernel_on_odd_sizes(cuda, oracle, npix, knum):
    """k_bwd_runs (D = 4: four consecutive sorted hits per lane): hit counts that are no multiple of four (the last lanes load one by
    one), k = 1 (runs of one or two hits: heads, tails and whole runs inside a lane), against fp64 autograd and against the round-5
    kernel (DEFTET_RAST_BWD=sorted is read once per process: the comparison runs in a child process)"""
    from deftet_amd.render import deftet_sparse_render
    fz, fxy, ff = projected_grid(6)
    pix, rngs = pixel_grid(npix)
    pix = pix * 0.6
    tp, tr, tz = (torch.from_numpy(x).to(cuda) for x in (pix, rngs, fz))
    txy = torch.from_numpy(fxy).to(cuda).requires_grad_(True)
    tff = torch.from_numpy(ff).to(cuda).requires_grad_(True)
    feat, face = deftet_sparse_render(tp, tr, tz, txy, tff, knum=knum)
    assert (pix.shape[1] * knum) % 4 != 0 or knum == 64
    go = torch.rand(feat.shape, device=cuda, generator=torch.Generator(device=cuda).manual_seed(7))
    gxy, gff = torch.autograd.grad(feat, (txy, tff), go)
    xy64 = txy.detach().double().requires_grad_(True)
    ff64 = tff.detach().double().requires_grad_(True)
    feat64 = oracle.sparse_render_torch(tp.double(), xy64, ff64, face)
    wxy, wff = torch.autograd.grad(feat64, (xy64, ff64), go.double())
    from tests.tol import check_close
    for nm, got, want in (("xy", gxy, wxy), ("feat", gff, wff)):
        assert want.abs().max().item() > 0
        check_close("A12 grad_%s (k_bwd_runs), %dx%d pixels k=%d vs fp64 autograd" % (nm, npix, npix, knum), got, want, 6e-7, elem_rel=3e-5)
    # twice the same bits
    gxy2, gff2 = torch.autograd.grad(deftet_sparse_render(tp, tr, tz, txy, tff, knum=knum)[0], (txy, tff), go)
    assert torch.equal(gxy, gxy2) and torch.equal(gff, gff2)


def test_backward_kernels_agree_in_a_child_process(cuda):
    """DEFTET_RAST_BWD=sorted (the round-5 reduction, one hit per lane) and the default (k_bwd_runs) on the same inputs: the same
    gradients up to the order of the fp32 additions"""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from tests.test_raster_gpu import projected_grid, pixel_grid\n"
        "from deftet_amd.render import deftet_sparse_render\n"
        "dev = torch.device('cuda:0')\n"
        "out = {}\n"
        "for case, (res, npix, knum, zoom) in enumerate([(8, 61, 33, 0.6), (4, 97, 7, 0.5), (10, 40, 64, 0.25), (6, 128, 2, 0.7), (12, 50, 16, 0.9)]):\n"
        "    fz, fxy, ff = projected_grid(res)\n"
        "    pix, rngs = pixel_grid(npix)\n"
        "    t = [torch.from_numpy(x).to(dev) for x in (pix * zoom, rngs, fz, fxy, ff)]\n"
        "    t[3].requires_grad_(True); t[4].requires_grad_(True)\n"
        "    feat, face = deftet_sparse_render(*t, knum=knum)\n"
        "    go = torch.rand(feat.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(case))\n"
        "    g = torch.autograd.grad(feat, (t[3], t[4]), go)\n"
        "    out['gxy%%d' %% case] = g[0].cpu().numpy(); out['gff%%d' %% case] = g[1].cpu().numpy()\n"
        "np.savez(sys.argv[1], **out)\n" % root)
    outs = []
    with tempfile.TemporaryDirectory() as d:
        for mode in ("sorted", "runs"):
            f = os.path.join(d, mode + ".npz")
            env = dict(os.environ, DEFTET_RAST_BWD=mode)
            r = subprocess.run([sys.executable, "-c", code, f], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout + r.stderr
            outs.append(dict(np.load(f)))
    assert len(outs[0]) == 10
    for k in sorted(outs[0]):                                           # five scenes (coarse / fine faces, k from 2 to 64, zoomed in and out)
        a, b = outs[0][k], outs[1][k]
        assert np.abs(b).max() > 0
        assert np.abs(a - b).max() <= 2e-6 * np.abs(a).max(), (k, np.abs(a - b).max(), np.abs(a).max())
        assert ((a == 0) == (b == 0)).all()                             # faces without hits: exactly zero in both


def test_backward_face_with_thousands_of_hits(cuda, oracle):
    """two big triangles covering a 96 x 96 pixel grid: 9,216 hits per face, i.e. runs of sorted hits that span nine
    blocks of the backward kernel (partial sums meet through float atomics), next to small faces"""
    from deftet_amd.render import deftet_sparse_render
    rng = np.random.default_rng(5)
    pix, rngs = pixel_grid(96)
    big = np.array([[[-3000.0, -3000.0], [5000.0, -3000.0], [-3000.0, 5000.0]],
                    [[-2500.0, -3500.0], [6000.0, -2000.0], [-3500.0, 6000.0]]], np.float32)
    small = (rng.uniform(-900, 900, (300, 1, 2)) + rng.uniform(-60, 60, (300, 3, 2))).astype(np.float32)
    fxy = np.concatenate([small[:150], big, small[150:]], 0)[None]
    F = fxy.shape[1]
    fz = rng.uniform(-900, -100, (1, F, 3)).astype(np.float32)
    ff = rng.random((1, F, 3, 4)).astype(np.float32)
    tp, tr, tz = (torch.from_numpy(x).to(cuda) for x in (pix, rngs, fz))
    txy = torch.from_numpy(fxy).to(cuda).requires_grad_(True)
    tff = torch.from_numpy(ff).to(cuda).requires_grad_(True)
    feat, face = deftet_sparse_render(tp, tr, tz, txy, tff, knum=16)
    assert (face == 150).sum().item() == 96 * 96 and (face == 151).sum().item() == 96 * 96
    go = torch.rand(feat.shape, device=cuda, generator=torch.Generator(device=cuda).manual_seed(4))
    gxy, gff = torch.autograd.grad(feat, (txy, tff), go)
    xy64 = txy.detach().double().requires_grad_(True)
    ff64 = tff.detach().double().requires_grad_(True)
    feat64 = oracle.sparse_render_torch(tp.double(), xy64, ff64, face)
    wxy, wff = torch.autograd.grad(feat64, (xy64, ff64), go.double())
    from tests.tol import check_close
    for nm, got, want in (("xy", gxy, wxy), ("feat", gff, wff)):
        for f in (150, 151):
            check_close("A12 grad_%s of a face with 9,216 hits vs fp64 autograd" % nm, got[0, f], want[0, f], 1e-4)
        check_close("A12 grad_%s, 96x96 pixels, two faces with 9,216 hits vs fp64 autograd" % nm, got, want, 2e-4)


def test_backward_baseline_config_fp64(cuda, oracle):
    """BASELINE configs[4] backward (15 M hits, faces with more than 64 hits straddle waves): fp64 autograd of the
    same interpolation on the same face indices, evaluated on the GPU."""
    from deftet_amd.render import deftet_sparse_render
    fz, fxy, ff = projected_grid(70)
    pix, rngs = pixel_grid(512)
    tp, tr, tz = (torch.from_numpy(x).to(cuda) for x in (pix, rngs, fz))
    txy = torch.from_numpy(fxy).to(cuda).requires_grad_(True)
    tff = torch.from_numpy(ff).to(cuda).requires_grad_(True)
    feat, face = deftet_sparse_render(tp, tr, tz, txy, tff, knum=64)
    g = torch.Generator(device=cuda).manual_seed(1)
    go = torch.rand(feat.shape, device=cuda, generator=g)
    gxy, gff = torch.autograd.grad(feat, (txy, tff), go)
    gxy2, gff2 = torch.autograd.grad(deftet_sparse_render(tp, tr, tz, txy, tff, knum=64)[0], (txy, tff), go)
    xy64 = txy.detach().double().requires_grad_(True)
    ff64 = tff.detach().double().requires_grad_(True)
    feat64 = oracle.sparse_render_torch(tp.double(), xy64, ff64, face)
    assert torch.allclose(feat64.float(), feat.detach(), rtol=1e-4, atol=1e-4)
    wxy, wff = torch.autograd.grad(feat64, (xy64, ff64), go.double())
    for got, again, want in ((gxy, gxy2, wxy), (gff, gff2, wff)):
        # per-face error relative to that face's own gradient magnitude (sums of up to a few hundred hits)
        mag = want.abs().reshape(want.shape[1], -1).max(-1).values.clamp(min=1e-30)
        err = (got.double() - want).abs().reshape(want.shape[1], -1).max(-1).values
        big = mag > 1e-3 * mag.max()
        assert (err[big] / mag[big]).max().item() < 2e-3
        from tests.tol import check_close
        # (xy: a face's gradient is the sum of up to a few hundred per-hit terms of mixed sign, each ~1/extent large: fp32
        # accumulation error relative to the LARGEST entry is sqrt(hits) * u * sum|terms| / max — measured 3.6e-5; feat: 4e-7)
        check_close("A12 grad (%s), configs[4] 512x512 k64 vs fp64 autograd" % ("xy" if got is gxy else "feat"), got, want,
                    8e-5 if got is gxy else 1e-6)
        # run-to-run: only faces whose hits straddle three or more waves may differ, and only by rounding
        assert (again.double() - got.double()).abs().max().item() <= 1e-5 * want.abs().max().item()
    hit = torch.zeros(fxy.shape[1], dtype=torch.bool, device=cuda)
    hit[face[face >= 0]] = True
    assert (gxy[0][~hit] == 0).all() and (gff[0][~hit] == 0).all()
    assert (gff[0][hit].abs().reshape(int(hit.sum()), -1).max(-1).values > 0).all()


def test_baseline_config_properties(cuda):
    """BASELINE configs[4]: 512x512 rays, k=64, unique faces of the res=70 grid."""
    fz, fxy, ff = projected_grid(70)
    pix, rngs = pixel_grid(512)
    feat, face = run(pix, rngs, fz, fxy, ff, 64, cuda)
    F = fxy.shape[1]
    assert F > 500000
    assert face.shape == (1, 512 * 512, 64) and feat.shape == (1, 512 * 512, 64, 4)
    valid = face >= 0
    assert ((face >= -1) & (face < F)).all()
    n = valid.sum(-1)
    assert n.max().item() == 64 and (n == 0).float().mean().item() > 0.01         # rays through the grid saturate, corners miss
    # valid slots form a prefix
    assert (valid[..., 1:] <= valid[..., :-1]).all()
    assert (feat[~valid] == 0).all()
    fv = feat[valid]
    assert fv.min().item() >= -1e-5 and fv.max().item() <= 1 + 1e-5             # convex combinations of features in [0,1)
    # recompute depth of every recorded hit from the face and the pixel; must be sorted nearest-first
    sub = slice(0, 512 * 512, 7)
    fsub = face[0, sub]
    vs = fsub >= 0
    fi = fsub.clamp(min=0)
    txy = torch.from_numpy(fxy).to(cuda)[0][fi]
    tz = torch.from_numpy(fz).to(cuda)[0][fi]
    p = torch.from_numpy(pix).to(cuda)[0, sub][:, None, :]
    a, b, c = txy[..., 0, :], txy[..., 1, :], txy[..., 2, :]
    m, pp, nn, q = b[..., 0] - a[..., 0], b[..., 1] - a[..., 1], c[..., 0] - a[..., 0], c[..., 1] - a[..., 1]
    s, t = p[..., 0] - a[..., 0], p[..., 1] - a[..., 1]
    den = (m * q - nn * pp) + 1e-8
    w1, w2 = (s * q - nn * t) / den, (m * t - s * pp) / den
    w0 = 1 - w1 - w2
    assert (torch.stack([w0, w1, w2], -1)[vs] >= 0).all()
    z = (w0 * tz[..., 0] + w1 * tz[..., 1]) + w2 * tz[..., 2]
    z = torch.where(vs, z, torch.full_like(z, -1e30))
    assert (z[:, 1:] <= z[:, :-1]).all()
    assert (z[vs] <= 0).all() and (z[vs] >= -1000).all()


def test_default_policy_warns_once_when_pixels_fill_up(cuda):
    """policy=None (round 6): the call renders NEAREST and says ONCE that pixels came back full — which faces a saturated pixel keeps
    is the one thing Kaolin's absence leaves open; an explicit policy never warns; saturated_pixels counts where the two differ."""
    import warnings
    from deftet_amd import grids, hip_ops
    import importlib
    mod = importlib.import_module("deftet_amd.render.deftet_sparse_render")   # (the package exports the FUNCTION under the same name)
    from deftet_amd.render.deftet_sparse_render import deftet_sparse_render, saturated_pixels, NEAREST, FIRST
    verts, tets = grids.kuhn_grid(8)
    f3 = hip_ops.tet_to_face(tets, verts.shape[0], cuda, with_boundary=True)[0].cpu().numpy()
    fz, fxy, ff = grids.project_faces(verts, f3, seed=0)
    pix, rngs = grids.pixel_grid(32)
    t = [torch.from_numpy(x).to(cuda) for x in (pix, rngs, fz, fxy, ff)]
    k = 4                                                            # far fewer slots than covering faces
    mod._warned_saturation, mod._default_calls = False, 0
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        a = deftet_sparse_render(*t, knum=k)
        b = deftet_sparse_render(*t, knum=k)
        c = deftet_sparse_render(*t, knum=k, policy=NEAREST)
        d = deftet_sparse_render(*t, knum=k, policy=FIRST)
    msgs = [w for w in rec if "saturation" in str(w.message)]
    assert len(msgs) == 1 and "policy=" in str(msgs[0].message)
    assert torch.equal(a[1], c[1]) and torch.equal(b[1], c[1])     # the default IS NEAREST
    n_sat = saturated_pixels(*t, knum=k)
    differ = int((c[1] != d[1]).any(-1).sum())
    assert 0 < differ <= n_sat <= int((c[1][..., k - 1] >= 0).sum())
    mod._warned_saturation, mod._default_calls = False, 0
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        deftet_sparse_render(*t, knum=4000)                          # never fills up: silent
    assert not [w for w in rec if "saturation" in str(w.message)]


def test_sliver_faces_have_certified_boxes(cuda, oracle):
    """Edge-on faces (area 2^-8 .. 2^-17 of their extent squared) are pruned per pixel chunk by a box enlarged by a quarter of
    their extent (raster.hip, face_box; round 6).  Pixels are placed where the contract is most likely to accept outside the
    face's own box — on the faces' lines, beyond their ends, a fraction of their width off — at several coordinate magnitudes;
    slivers thinner than 2^-16, collinear and tiny ones keep the unbounded path.  Bit-exact against the brute-force oracle."""
    rng = np.random.default_rng(11)
    F = 260
    fxy = np.zeros((1, F, 3, 2), np.float32)
    lines = []
    for i in range(F):
        k = rng.integers(6, 20)                                        # area ratio ~ 2^-k: regular, sliver and degenerate faces
        off = [0.0, 5.0, 300.0, 20000.0][i % 4]
        L = rng.uniform(0.05, 40.0) * (1.0 if off < 1e3 else 50.0)
        th = rng.uniform(0, 2 * np.pi)
        d, dp = np.array([np.cos(th), np.sin(th)]), np.array([-np.sin(th), np.cos(th)])
        a = rng.uniform(-1, 1, 2) * off
        h = L * 2.0 ** (-float(k)) * rng.uniform(0.6, 1.6)
        fxy[0, i, 0], fxy[0, i, 1], fxy[0, i, 2] = a, a + L * d, a + rng.uniform(0.2, 1.0) * L * d + h * dp
        lines.append((a, d, dp, L, h))
    fxy[0, 250] = fxy[0, 3][[0, 0, 2]]                                 # two equal corners
    fxy[0, 251, 2] = (fxy[0, 251, 0] + fxy[0, 251, 1]) / 2             # collinear
    fz = rng.uniform(-5, -1, (1, F, 3)).astype(np.float32)
    ff = rng.random((1, F, 3, 3)).astype(np.float32)
    pts = []
    for i in range(F):
        a, d, dp, L, h = lines[i]
        lam = rng.uniform(-1.5, 2.5, 24) * L
        mu = rng.normal(0, 1, 24) * h * rng.choice([0.1, 1.0, 10.0, 100.0], 24)
        pts.append(a[None] + lam[:, None] * d[None] + mu[:, None] * dp[None])
        pts.append(fxy[0, i].astype(np.float64))                       # the corners themselves
    pix = np.concatenate(pts)[None].astype(np.float32)
    P = pix.shape[1]
    rngs = np.tile(np.array([-1000.0, 0.0], np.float32), (1, P, 1))
    for knum, policy in ((6, NEAREST), (6, FIRST), (64, NEAREST)):
        wf, wface, ww = oracle.sparse_render_fwd(pix, rngs, fz, fxy, ff, knum=knum, policy=policy)
        feat, face = run(pix, rngs, fz, fxy, ff, knum, cuda, policy=policy)
        assert np.array_equal(face.cpu().numpy(), wface), (knum, policy)
        assert np.array_equal(feat.cpu().numpy(), wf, equal_nan=True)
    assert (wface >= 0).sum() > 2000                                  # the pixels do land on the faces
