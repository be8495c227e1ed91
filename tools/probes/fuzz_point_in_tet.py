"""Fuzz: binned point-in-tet variants and both backwards against the independent brute-force HIP kernel
on random sizes / tet soups / query patterns.   python tools/probes/fuzz_point_in_tet.py [seconds]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deftet_amd import hip_ops

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
g = torch.Generator(device=dev).manual_seed(7)
t_end = time.time() + budget
n = 0
csr_cache = {}
while time.time() < t_end:
    B = int(rng.integers(1, 5)); T = int(rng.choice([1, 2, 5, 63, 64, 65, 300, 1000, 3000, 20000])); Q = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 6000, 30000]))
    kind = rng.integers(0, 8)
    scale = float(rng.choice([1e-3, 1.0, 1.0, 1.0, 50.0]))
    size = float(rng.choice([0.02, 0.1, 0.4, 1.5]))
    if kind >= 5:
        # a COHERENT mesh (what the wave-staged traversal groups): jittered Kuhn grid in cube-major order, sometimes with
        # stretches of it reversed or rotated, jitter from mild to inverting; kinds 6 / 7 make the query set dense
        # (several staged chunks per footprint, rows that do not fit a chunk)
        from deftet_amd import grids
        res = int(rng.choice([2, 4, 6, 10, 16, 24]))
        verts, tets = grids.kuhn_grid(res)
        T = tets.shape[0]
        amp = float(rng.choice([0.0, 0.1, 0.3, 0.9]))
        v = torch.from_numpy(verts.astype(np.float32)).to(dev)[None].repeat(B, 1, 1) - 0.5
        v = v + amp / res * (torch.rand(v.shape, device=dev, generator=g) - 0.5)
        idx = torch.from_numpy(tets.astype(np.int64)).to(dev)
        if rng.random() < 0.5:
            k0 = int(rng.integers(0, T)); k1 = int(rng.integers(k0, T + 1))
            idx = torch.cat([idx[:k0], idx[k0:k1].flip(0), idx[k1:]])
        if rng.random() < 0.3:
            idx = idx.roll(int(rng.integers(0, T)), 0)
        tet = (v[:, idx] * scale).contiguous()
    else:
        c = torch.rand(B, T, 1, 3, device=dev, generator=g) - 0.5
        tet = (c + size * (torch.rand(B, T, 4, 3, device=dev, generator=g) - 0.5)) * scale
    if kind == 1:                                        # degenerate / non-finite tets mixed in
        m = torch.rand(B, T, device=dev, generator=g)
        tet[m < 0.05] = tet[m < 0.05][:, :1].expand(-1, 4, -1)            # collapsed
        tet[(m > 0.05) & (m < 0.07)] = float("nan")
        tet[(m > 0.07) & (m < 0.09)] *= 1e7
    pts = (torch.rand(B, Q, 3, device=dev, generator=g) - 0.5) * scale * 1.1
    if kind == 2 or kind == 6:
        pts = pts * 1e-3 + 0.1 * scale                   # one cell
    if kind == 7:
        pts[:, : Q // 2] = pts[:, : Q // 2] * 0.05 + 0.2 * scale    # half of them in a small box: dense rows next to sparse ones
    if kind == 3:
        pts[..., int(rng.integers(0, 3))] = 0.03 * scale  # a plane
    if kind == 4:
        k = max(1, Q // 20)
        pts[:, :k] = float("nan"); pts[:, k:2 * k] = 3e6 * scale
    ref = hip_ops.point_in_tet(tet, pts, algo=1)
    for algo in (0, 2, 3, 4, 5):
        # a random traversal order and a random query box now and then: neither may change a result
        order = None
        if T > 1 and rng.random() < 0.3:
            order = torch.from_numpy(rng.permutation(T).astype("int32")).to(dev) if rng.random() < 0.5 else hip_ops.tet_spatial_order(tet[0])
        box = None
        if rng.random() < 0.4:
            c0 = (torch.rand(B, 3, device=dev, generator=g) - 0.5) * scale
            half = torch.rand(B, 3, device=dev, generator=g) * scale * float(rng.choice([0.01, 0.3, 0.7, 2.0]))
            box = torch.cat([c0 - half, c0 + half], 1).contiguous()
        miss = torch.full((B,), -1, device=dev, dtype=torch.int32) if box is not None and algo != 1 else None
        got = hip_ops.point_in_tet(tet, pts, algo=algo, order=order, query_box=box, query_box_misses=miss)
        if miss is not None:                                 # the miss count the box tracker relies on: regular queries outside the enlarged box
            lo, hi = box[:, :3], box[:, 3:]
            e = (hi - lo) * (1.0 / 32.0)
            glo, ghi = torch.clamp(lo - e, min=-1048576.0), torch.clamp(hi + e, max=1048576.0)
            want = ((pts.abs() <= 1048576.0).all(-1) & ~((pts >= glo[:, None]) & (pts <= ghi[:, None])).all(-1)).sum(1).to(torch.int32)
            if not torch.equal(miss, want):
                print("MISS COUNT algo=%d B=%d T=%d Q=%d kind=%d: %s vs %s" % (algo, B, T, Q, kind, miss.tolist(), want.tolist()), flush=True)
                sys.exit(1)
        if not torch.equal(got, ref):
            bad = (got != ref).nonzero()[0].tolist()
            print("MISMATCH algo=%d B=%d T=%d Q=%d kind=%d scale=%g size=%g at %s: %s vs %s" % (algo, B, T, Q, kind, scale, size, bad, got[tuple(bad)].item(), ref[tuple(bad)].item()), flush=True)
            sys.exit(1)
    pred = torch.rand(B, T, device=dev, generator=g)
    falgo = int(rng.choice([0, 2, 3, 4, 5]))
    fbox = None
    if rng.random() < 0.4:
        c0 = (torch.rand(B, 3, device=dev, generator=g) - 0.5) * scale
        half = torch.rand(B, 3, device=dev, generator=g) * scale * float(rng.choice([0.3, 0.7, 2.0]))
        fbox = torch.cat([c0 - half, c0 + half], 1).contiguous()
    cond, w, occ, hits = hip_ops.point_in_tet(tet, pts, want_bary=True, pred_bxt=pred, want_hits=True, algo=falgo, query_box=fbox)
    assert torch.equal(cond, ref)
    gw = torch.randn(B, Q, 4, device=dev, generator=g); go = torch.randn(B, Q, device=dev, generator=g)
    a = hip_ops.point_in_tet_bwd(tet, pts, cond, gw, grad_occ=go, hits=hits)
    b = hip_ops.point_in_tet_bwd(tet, pts, cond, gw, grad_occ=go)
    for name, x, y in (("grad_tet", a[0], b[0]), ("grad_pred", a[2], b[2])):
        f = torch.isfinite(x) & torch.isfinite(y)
        if f.any():
            err = ((x - y)[f]).abs().max().item(); ref_mag = y[f].abs().max().item()
            if err > 2e-3 * max(ref_mag, 1e-30):
                d = torch.where(f, (x - y).abs(), torch.zeros_like(x))
                at = [int(v) for v in torch.unravel_index(d.argmax(), d.shape)]
                words = hits[2 * B * T:2 * B * T + 3 * ((B + 63) // 64 * 64)].view(3, -1)[:, :B].tolist()
                qs = (cond[at[0], :, 0] == at[1]).nonzero().flatten().tolist()
                nU = words[0][at[0]]
                l0 = 2 * B * T + 3 * ((B + 63) // 64 * 64)
                ul = hits[l0:l0 + B * Q].view(B, Q)[at[0], :nU].tolist()
                print("queries won by that tet: %s (coordinates %s); in the uncovered list: %s" % (
                    qs, pts[at[0], qs].tolist(), [q in ul for q in qs]), flush=True)
                print("forward algo %d; scale %g size %g" % (falgo, scale, size), flush=True)
                print("BACKWARD MISMATCH %s B=%d T=%d Q=%d kind=%d: err %g of %g at %s: hits-path %g vs list-path %g; "
                      "uncovered/ticket/irregular-query words %s; record of that tet %s" % (
                          name, B, T, Q, kind, err, ref_mag, at, x[tuple(at)].item(), y[tuple(at)].item(), words,
                          hits[:2 * B * T].view(B, T, 2)[at[0], at[1]].tolist()), flush=True)
                sys.exit(1)
    # the fused backward onto the vertices (compacted rows + mask words, masked gather) with every tet owning its four vertices:
    # grad_pos is then grad_tet itself — bit for bit on the record path, to round-off where the per-tet lists take over
    csr = csr_cache.get(T)
    if csr is None:
        csr = csr_cache[T] = hip_ops.tet_vertex_csr(torch.arange(4 * T, device=dev, dtype=torch.int64).view(T, 4), 4 * T)
    fv = hip_ops.point_in_tet_bwd_to_vertices(tet, pts, cond, gw, csr, 4 * T, grad_occ=go, hits=hits)
    x, y = fv[0].view(B, T, 4, 3), a[0]
    f = torch.isfinite(x) & torch.isfinite(y)
    ok = bool((torch.isfinite(x) == torch.isfinite(y)).all())
    if ok and f.any():
        ok = bool((x[f] == y[f]).all()) if Q <= 2 * T else ((x - y)[f]).abs().max().item() <= 2e-3 * max(y[f].abs().max().item(), 1e-30)
    if ok and Q <= 2 * T:
        fp = torch.isfinite(fv[2]) & torch.isfinite(a[2])
        ok = bool((fv[2][fp] == a[2][fp]).all())
    if not ok:
        print("TO-VERTICES MISMATCH B=%d T=%d Q=%d kind=%d falgo=%d scale=%g size=%g" % (B, T, Q, kind, falgo, scale, size), flush=True)
        sys.exit(1)
    n += 1
print("fuzz ok: %d random cases" % n, flush=True)
