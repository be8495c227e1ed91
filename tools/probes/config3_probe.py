"""BASELINE configs[3]: res=100 (T=750,000), Q=200,000, B=64 shapes on ONE GPU — does the fwd+bwd step run
at that size, how long does it take, and do its size-independent properties hold?"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deftet_amd import grids, hip_ops

dev = torch.device("cuda:0")
res, Q, B = 100, 200000, 64
verts, tets = grids.kuhn_grid(res)
idx = torch.from_numpy(tets.astype(np.int64)).to(dev)
base = torch.from_numpy((verts - 0.5).astype(np.float32)).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
h = 2.0 / res
interior = ((base.abs() < 0.5 - 1e-6).all(1, keepdim=True)).float()
pos = base[None] + interior[None] * (torch.rand(B, base.shape[0], 3, device=dev, generator=g) * 0.2 - 0.1) * h
tet = hip_ops.tet_gather(pos, idx)                                  # [64, 750000, 4, 3] = 2.3 GB
pts = 1.05 * (torch.rand(B, Q, 3, device=dev, generator=g) - 0.5)
T = tet.shape[1]
pred = torch.rand(B, T, device=dev, generator=g); gw = torch.randn(B, Q, 4, device=dev, generator=g); go = torch.randn(B, Q, device=dev, generator=g)


def step():
    cond, w, occ, hits = hip_ops.point_in_tet(tet, pts, want_bary=True, pred_bxt=pred, want_hits=True)
    g_tet, _, g_pred = hip_ops.point_in_tet_bwd(tet, pts, cond, gw, grad_occ=go, hits=hits)
    return cond, w, occ, g_tet, g_pred


for _ in range(2):
    out = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
cond, w, occ, g_tet, g_pred = out
hit = cond[..., 0] >= 0
sel = torch.gather(tet[:4], 1, cond[:4, :, 0].clamp(min=0).long()[:, :, None, None].expand(-1, -1, 4, 3))
rec = (w[:4, :, :, None] * sel).sum(2)
print("configs[3]: B=%d T=%d Q=%d: %.3f ms per fwd+bwd step = %.3e M nominal tests/s; miss rate %.3f; "
      "max |sum w_i v_i - p| over hits = %.2e; grad finite: %s; peak memory %.1f GB"
      % (B, T, Q, dt * 1e3, B * T * Q / dt / 1e6, 1 - hit.float().mean().item(), (rec - pts[:4])[hit[:4]].abs().max().item(),
         bool(torch.isfinite(g_tet).all() and torch.isfinite(g_pred).all()), torch.cuda.max_memory_allocated() / 1e9), flush=True)
