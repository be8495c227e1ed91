"""Does a whole fwd+bwd step capture into a HIP graph (torch.cuda.CUDAGraph) and what does replay cost?
Small shapes are launch-bound (11 launches), so the graph should help there."""
import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deftet_amd import grids, hip_ops

dev = torch.device("cuda:0")
for res, nq, B in [(20, 10000, 1), (40, 50000, 8), (70, 100000, 8)]:
    tet, pts, _, _ = grids.make_case(res, nq, B)
    t, p = torch.from_numpy(tet).to(dev), torch.from_numpy(pts).to(dev)
    T = t.shape[1]
    pred = torch.rand(B, T, device=dev); gw = torch.randn(B, nq, 4, device=dev); go = torch.randn(B, nq, device=dev)

    def step():
        cond, w, occ, hits = hip_ops.point_in_tet(t, p, want_bary=True, pred_bxt=pred, want_hits=True)
        g_tet, _, g_pred = hip_ops.point_in_tet_bwd(t, p, cond, gw, grad_occ=go, hits=hits)
        loss = hip_ops.rowdot(w, gw, occ, go)
        return cond, g_tet, g_pred, loss

    for _ in range(3):
        ref = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 50
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = step()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    ok = [bool(torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))) for a, b in zip(out, ref)]
    ok2 = [bool(torch.equal(a, b)) for a, b in zip(step(), step())]
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 50
    print("res=%d Q=%d B=%d: eager %.1f us, graph replay %.1f us, graph==eager %s, eager==eager %s" % (res, nq, B, eager * 1e6, graph * 1e6, ok, ok2), flush=True)
