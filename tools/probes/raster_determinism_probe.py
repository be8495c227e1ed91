"""deftet_sparse_render fwd + bwd at BASELINE configs[4], repeated on the same inputs: are image, depth and the three gradients the same
bits every time?  python tools/probes/raster_determinism_probe.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
dev = torch.device("cuda:0")
for policy in (None, "1"):
    if policy:
        os.environ["DEFTET_BENCH_RASTER_POLICY"] = policy
    wl = bench.RasterWorkload(bench.CONFIGS[4], 0, dev, 1)
    ref, nd = None, {}
    for it in range(5):
        wl.step(it)
        torch.cuda.synchronize()
        cur = [t.detach().clone() for t in wl.last]
        if ref is None:
            ref = cur
        else:
            for k, (a, b) in enumerate(zip(cur, ref)):
                if not torch.equal(a.view(torch.int32), b.view(torch.int32)):
                    nd[k] = nd.get(k, 0) + 1
    print("saturation policy %s: runs (of 4) that differ from the first, per output %s: %s" % ({None: "nearest-k", "1": "first-k"}[policy], ["features", "face index", "grad_xy", "grad_features"], nd or "none"))
