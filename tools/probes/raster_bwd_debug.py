"""Debug: rasterizer backward vs a torch index_add reference for grad_face_features."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_raster_gpu import pixel_grid, projected_grid
from deftet_amd.render import deftet_sparse_render
from oracle import oracle
dev = torch.device("cuda:0")
res, npx, knum = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (6, 24, 48)
fz, fxy, ff = projected_grid(res)
pix, rngs = pixel_grid(npx)
pix = pix * 0.6
tp, tr, tz = (torch.from_numpy(x).to(dev) for x in (pix, rngs, fz))
txy = torch.from_numpy(fxy).to(dev).requires_grad_(True)
tff = torch.from_numpy(ff).to(dev).requires_grad_(True)
feat, face = deftet_sparse_render(tp, tr, tz, txy, tff, knum=knum)
go = torch.rand(feat.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
gxy, gff = torch.autograd.grad(feat, (txy, tff), go)
xy64 = txy.detach().double().requires_grad_(True)
ff64 = tff.detach().double().requires_grad_(True)
feat64 = oracle.sparse_render_torch(tp.double(), xy64, ff64, face)
wxy, wff = torch.autograd.grad(feat64, (xy64, ff64), go.double())
F = fxy.shape[1]
cnt = torch.bincount(face[face >= 0].flatten(), minlength=F)
for name, got, want in (("gxy", gxy, wxy), ("gff", gff, wff)):
    err = (got.double() - want).abs().reshape(F, -1).max(-1).values
    mag = want.abs().reshape(F, -1).max(-1).values
    bad = (err > 1e-4 * want.abs().max()).nonzero().flatten()
    print(name, "max err", err.max().item(), "scale", want.abs().max().item(), "bad faces", bad.numel(), "of", int((cnt > 0).sum()))
    for f in bad[:8].tolist():
        print("  face", f, "hits", int(cnt[f]), "got", got[0, f].flatten()[:4].tolist(), "want", want[0, f].flatten()[:4].tolist())
feat, face = deftet_sparse_render(tp, tr, tz, txy, tff, knum=knum)
go1 = torch.ones_like(feat)
_, g1 = torch.autograd.grad(feat, (txy, tff), go1)
got_cnt = g1[0, :, :, 0].sum(-1)
keys = torch.where(face >= 0, face, torch.full_like(face, F)).flatten()
sk, order = torch.sort(keys, stable=True)
start = torch.searchsorted(sk, torch.arange(F + 1, device=dev))
nbad = 0
for f in range(F):
    c = int(cnt[f])
    if c and abs(got_cnt[f].item() - c) > 1e-3 * c:
        s, e = int(start[f]), int(start[f + 1])
        print("face", f, "cnt", c, "got", round(got_cnt[f].item(), 3), "sorted range", s, e, "lanes", s % 64, (e - 1) % 64, "waves", s // 64, (e - 1) // 64)
        nbad += 1
        if nbad > 25:
            break
