"""Rasterizer forward + backward at BASELINE configs[4] a few times (for rocprofv3 --kernel-trace --stats)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_raster_gpu import pixel_grid, projected_grid
from deftet_amd.render import deftet_sparse_render
dev = torch.device("cuda:0")
fz, fxy, ffe = projected_grid(70)
pix, rngs = pixel_grid(512)
t = [torch.from_numpy(x).to(dev) for x in (pix, rngs, fz, fxy, ffe)]
t[3].requires_grad_(True)
t[4].requires_grad_(True)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
feat, face = deftet_sparse_render(*t, knum=64)
go = torch.rand_like(feat)
for name, fn in (("fwd", lambda: deftet_sparse_render(*t, knum=64)),
                 ("bwd", lambda: torch.autograd.grad(feat, (t[3], t[4]), go, retain_graph=True))):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(name, "ms", round(e0.elapsed_time(e1) / reps, 3))
