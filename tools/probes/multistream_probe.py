"""Does spreading a rank's shapes over several HIP streams shorten the fwd+bwd step?  (SURVEY 8(e):
"Within a rank, shapes may be further spread over streams".)  Each stream runs the full operator
chain on its slice of the batch; small latency-bound kernels of one slice can overlap the big
kernels of another."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deftet_amd import grids, hip_ops

dev = torch.device("cuda:0")
res, nq, B = 70, 100000, 8
tet, pts, _, _ = grids.make_case(res, nq, B)
t, p = torch.from_numpy(tet).to(dev), torch.from_numpy(pts).to(dev)
T = t.shape[1]
pred = torch.rand(B, T, device=dev); gw = torch.randn(B, nq, 4, device=dev); go = torch.randn(B, nq, device=dev)


def chain(sl):
    cond, w, occ, hits = hip_ops.point_in_tet(t[sl], p[sl], want_bary=True, pred_bxt=pred[sl], want_hits=True)
    g_tet, _, g_pred = hip_ops.point_in_tet_bwd(t[sl], p[sl], cond, gw[sl], grad_occ=go[sl], hits=hits)
    return hip_ops.rowdot(w, gw[sl], occ, go[sl])


for ns in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    per = B // ns
    slices = [slice(k * per, (k + 1) * per) for k in range(ns)]

    def step():
        cur = torch.cuda.current_stream()
        outs = []
        for s, sl in zip(streams, slices):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(chain(sl))
        for s in streams:
            cur.wait_stream(s)
        return torch.cat(outs)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    print("streams=%d: %.1f us per step" % (ns, (time.perf_counter() - t0) / 30 * 1e6), flush=True)
