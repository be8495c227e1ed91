"""GPU kernels of ONE geometry step (bench.py --config 5) stage by stage (the stages of DefTet.forward_surface_align + the
occupancy query + the stand-in loss + backward): where the launches come from.   python tools/probes/geometry_trace.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from deftet_amd import surface_losses
from deftet_amd.layers.DefTet.check_condition_tetrahedron_base.utils import point_in_tet_occ

wl = bench.make_workload(5, 0, torch.device("cuda:0"), 1)
for i in range(3):
    wl.step(i)
torch.cuda.synchronize()
m, pos, pred, gt = wl.m, wl.pos, wl.pred, wl.gt
idxB, f3, t2, gt_verts, gt_faces, pts, inv_v = wl.args
B = pos.shape[0]
m.inverse_v = inv_v


def stage(name, fn):
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        out = fn()
        torch.cuda.synchronize()
    ks = [e for e in prof.events() if e.device_type != torch.autograd.DeviceType.CPU]
    ks.sort(key=lambda e: e.time_range.start)
    tot = sum((e.device_time if hasattr(e, "device_time") else e.cuda_time) for e in ks)
    print("== %s: %d launches, %.1f us" % (name, len(ks), tot))
    for e in ks:
        print("     %-90s %7.1f" % (e.name[:90], e.device_time if hasattr(e, "device_time") else e.cuda_time))
    return out


pos.grad = None; pred.grad = None
tet = stage("gather_tet_pos", lambda: m.gather_tet_pos(pos, idxB))
occ_c = stage("check_tet_inside_sdfs", lambda: m.check_tet_inside_sdfs(tet, ([gt_verts[None]] * B, [[gt_faces]] * B)))
boundary = stage("get_boundary_index", lambda: m.get_boundary_index(f3, t2, occ_c.squeeze(dim=-1)))
en = stage("energies", lambda: m.energies(tet, inv_v))
terms = stage("surface_terms_batched", lambda: surface_losses.surface_terms_batched(pos, boundary, gt, per_face=20, stacked=True))
tm = stage("terms.mean + tuple", lambda: terms.mean(1, keepdim=True))
cwo = stage("occupancy_query", lambda: m.occupancy_query(pos, idxB, pts, pred, tet_bxfx4x3=tet))     # as step_demo.run_full_step
cond, w, occ = cwo
vvar, amips, edge = en
sc, sa, sn = tm
import step_demo
loss = stage("stand-in loss", lambda: step_demo.standin_loss(w, occ, amips, edge, vvar, sc, sa, sn))
stage("backward", lambda: loss.backward())
