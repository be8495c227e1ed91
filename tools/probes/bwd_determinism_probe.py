import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from deftet_amd import grids, hip_ops
cuda = torch.device('cuda:0')
B, Q = 2, 3100
"""Repeated forward + hit-record backward on the same inputs: how many of 19 repeats differ in any bit of grad_tet from the first
run (rounds 2-4: all of them wherever a tet overflowed its record; round 5: none).  python tools/probes/bwd_determinism_probe.py"""
for res, Qx in ((12, 2500), (20, 11000), (12, 600), (70, 100000), (40, 50000)):
    tet, pts, _, _ = grids.make_case(res, Qx, B, 0.1)
    pts = pts.copy(); pts[:, : Qx // 2] *= 0.5              # half of the queries in an eighth of the volume: those tets overflow their records
    t = torch.from_numpy(tet).to(cuda); p = torch.from_numpy(pts).to(cuda)
    g = torch.Generator(device=cuda).manual_seed(5)
    gw = torch.randn(B, Qx, 4, device=cuda, generator=g); go = torch.randn(B, Qx, device=cuda, generator=g)
    pred = torch.rand(B, t.shape[1], device=cuda, generator=g)
    ref = None
    nd = 0
    for it in range(20):
        cond, w, occ, hits = hip_ops.point_in_tet(t, p, want_bary=True, pred_bxt=pred, want_hits=True)
        gt, _, gp = hip_ops.point_in_tet_bwd(t, p, cond, gw, grad_occ=go, hits=hits)
        st = hip_ops.point_in_tet_stats(B, t.shape[1], Qx, 0, cuda)
        if ref is None: ref = (gt.clone(), gp.clone())
        else:
            d = (gt.view(torch.int32) != ref[0].view(torch.int32))
            if d.any():
                nd += 1
                if nd == 1:
                    idx = d.nonzero()[0].tolist()
                    print('  differs at', idx, gt[tuple(idx)].item(), ref[0][tuple(idx)].item(), 'tet hits count', int((cond[idx[0], :, 0] == idx[1]).sum()))
    print(res, Qx, 'T', t.shape[1], 'runs differing from the first:', nd, 'stats', st.tolist())
