"""Host-side cost of ONE geometry step (bench.py --config 5): wall time the CPU spends inside each stage while the GPU runs
asynchronously (no synchronisation added; get_boundary_index has its own read-back), and the step's wall time.  Tells whether
the step is bound by the host's launch rate or by the kernels.   python tools/probes/geometry_cpu_probe.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import step_demo
from deftet_amd import surface_losses

wl = bench.make_workload(5, 0, torch.device("cuda:0"), 1)
for i in range(5):
    wl.step(i)
torch.cuda.synchronize()
m, pos, pred, gt = wl.m, wl.pos, wl.pred, wl.gt
idxB, f3, t2, gt_verts, gt_faces, pts, inv_v = wl.args
B = pos.shape[0]
m.inverse_v = inv_v
acc = {}


def stage(name, fn):
    t0 = time.perf_counter()
    out = fn()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return out


N = 20
torch.cuda.synchronize()
w0 = time.perf_counter()
for _ in range(N):
    pos.grad = None
    pred.grad = None
    tet = stage("gather_tet_pos", lambda: m.gather_tet_pos(pos, idxB))
    occ_c = stage("check_tet_inside_sdfs", lambda: m.check_tet_inside_sdfs(tet, ([gt_verts[None]] * B, [[gt_faces]] * B)))
    boundary = stage("get_boundary_index (read-back: waits for the GPU)", lambda: m.get_boundary_index(f3, t2, occ_c.squeeze(dim=-1)))
    en = stage("energies", lambda: m.energies(tet, inv_v))
    terms = stage("surface_terms_batched", lambda: surface_losses.surface_terms_batched(pos, boundary, gt, per_face=20, stacked=True))
    tm = stage("terms.mean", lambda: terms.mean(1, keepdim=True))
    cond, w, occ = stage("occupancy_query", lambda: m.occupancy_query(pos, idxB, pts, pred, tet_bxfx4x3=tet))
    loss = stage("stand-in loss", lambda: step_demo.standin_loss(w, occ, en[1], en[2], en[0], tm[0], tm[1], tm[2]))
    stage("backward", lambda: loss.backward())
torch.cuda.synchronize()
wall = (time.perf_counter() - w0) / N
print(json.dumps({"wall_ms_per_step": round(wall * 1e3, 4), "host_ms_per_stage": {k: round(v / N * 1e3, 4) for k, v in acc.items()},
                  "host_ms_total": round(sum(acc.values()) / N * 1e3, 4)}))
