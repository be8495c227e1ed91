#!/usr/bin/env python
"""Secondary measurements for the SURVEY section 8 rows that are not the headline metric:
builders (A2-A6), surface ops (A8-A10) and the rasterizer (A12), each with its CPU comparator
(oracle/_ref = the reference's own native code where it exists, else the oracle port).
Prints one JSON line per op.   python tools/bench_ops.py [--quick]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deftet_amd import grids, hip_ops  # noqa: E402
from oracle import oracle as O  # noqa: E402

dev = torch.device("cuda:0")
quick = "--quick" in sys.argv


def gpu_time(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def gpu_time_events(fn, reps=20, warm=3):
    """Device time per call: HIP events around `reps` back-to-back calls (no host sync in between), so launch and
    synchronisation latency of a single call are not in the figure."""
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def roof(alg_bytes, seconds, bound):
    """Roofline fields of a secondary line: the HBM floor of the algorithmic bytes at 8 TB/s, and what fraction of the
    measured time that floor is; `bound` says what the kernel is limited by when that is not HBM."""
    floor = alg_bytes / 8e12
    return dict(algorithmic_mb=round(alg_bytes / 1e6, 2), hbm_floor_ms=round(floor * 1e3, 4), roofline_frac=round(floor / seconds, 4), bound=bound)


def cpu_time(fn):
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


def emit(**kw):
    print(json.dumps(kw), flush=True)


def main():
    res = 40 if quick else 70
    verts, tets = grids.kuhn_grid(res)
    n_point, T = verts.shape[0], tets.shape[0]
    tets_d = torch.from_numpy(tets).to(dev)
    ref = O.RefBuilders() if O.RefBuilders.available() else None
    for name, g, c_ref, c_port in [
        ("tet_adj_share", lambda: hip_ops.tet_adj_share(tets_d, n_point, dev), lambda: ref.tet_adj_share(tets, n_point), lambda: O.tet_adj_share(tets, n_point)),
        ("tet_face_adj", lambda: hip_ops.tet_face_adj(tets_d, n_point, dev), lambda: ref.tet_face_adj(tets, n_point), lambda: O.tet_face_adj(tets, n_point)),
        ("tet_point_adj", lambda: hip_ops.tet_point_adj(tets_d, n_point, dev), lambda: ref.tet_point_adj(tets, n_point), lambda: O.tet_point_adj(tets, n_point)),
        ("tet_to_face", lambda: hip_ops.tet_to_face(tets_d, n_point, dev), None, lambda: O.tet_to_face(tets, n_point)),
    ]:
        tg = gpu_time(g)
        kind, tc = ("reference", cpu_time(c_ref)) if (ref is not None and c_ref is not None) else ("port", cpu_time(c_port))
        emit(op=name, res=res, n_tet=T, gpu_ms=round(tg * 1e3, 3), cpu_ms=round(tc * 1e3, 1), cpu_kind=kind, cpu_cores=1,
             speedup=round(tc / tg, 1))

    # surface ops on a sphere surface: F boundary faces, P = 100k GT points, N = 20 F queries
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tests.test_surface_ops_gpu import sphere_surface
    face = sphere_surface(40 if quick else 70)
    F = face.shape[0]
    rng = np.random.default_rng(3000)
    d = rng.standard_normal((100000, 3))
    gt = (0.3 * d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    face_d, gt_d = torch.from_numpy(face).to(dev), torch.from_numpy(gt).to(dev)[None]
    nfb = torch.tensor([float(F)], device=dev)
    tg = gpu_time(lambda: hip_ops.face_edge_adj(face_d, 30))
    tgb = gpu_time(lambda: hip_ops.face_edge_adj(face_d, 30, brute=True))
    sub = min(F, 2000)
    tc = cpu_time(lambda: O.face_edge_adj(face[:sub], 30)) * (F / sub) ** 2
    emit(op="face_edge_adj", n_face=F, gpu_ms=round(tg * 1e3, 3), gpu_brute_ms=round(tgb * 1e3, 3), cpu_ms=round(tc * 1e3, 1), cpu_kind="port (extrapolated from %d faces, O(F^2))" % sub,
         cpu_cores=1, pairs_per_s=round(F * F / tg / 1e9, 2), unit="G face pairs/s")
    tg = gpu_time(lambda: hip_ops.tri_dist_fwd(gt_d, face_d[None], nfb), reps=3)
    tge = gpu_time_events(lambda: hip_ops.tri_dist_fwd(gt_d, face_d[None], nfb))
    subp = 500
    tc = cpu_time(lambda: O.tri_dist_fwd(gt[None, :subp], face[None], np.array([F], np.float32))) * (100000 / subp)
    tbr = gpu_time(lambda: hip_ops.tri_dist_fwd(gt_d, face_d[None], nfb, brute=True), reps=3)
    emit(op="tri_dist_fwd", n_face=F, n_point=100000, gpu_ms=round(tg * 1e3, 3), gpu_brute_ms=round(tbr * 1e3, 3), cpu_ms=round(tc * 1e3, 1),
         cpu_kind="port (extrapolated from %d points)" % subp, cpu_cores=1, pairs_per_s=round(F * 1e5 / tg / 1e9, 2), unit="G point-triangle pairs/s",
         device_ms=round(tge * 1e3, 4), **roof(100000 * (12 + 8) + F * 36, tge, "valu (nearest-triangle search: ~60 flop per candidate pair, tens of candidates per point)"))
    far_d = gt_d * 1.3                                                # early training: the cloud 30 % off the predicted surface
    tgf = gpu_time(lambda: hip_ops.tri_dist_fwd(far_d, face_d[None], nfb), reps=3)
    emit(op="tri_dist_fwd", points="30 % outside the surface (far path)", n_face=F, n_point=100000, gpu_ms=round(tgf * 1e3, 3),
         gpu_brute_ms=round(tbr * 1e3, 3), pairs_per_s=round(F * 1e5 / tgf / 1e9, 2), unit="G point-triangle pairs/s")
    dd, ff = hip_ops.tri_dist_fwd(gt_d, face_d[None], nfb)
    gg = torch.ones_like(dd)
    tg = gpu_time(lambda: hip_ops.tri_dist_bwd(gt_d, face_d[None], ff, gg))
    tge = gpu_time_events(lambda: hip_ops.tri_dist_bwd(gt_d, face_d[None], ff, gg))
    emit(op="tri_dist_bwd", n_face=F, n_point=100000, gpu_ms=round(tg * 1e3, 3), device_ms=round(tge * 1e3, 4),
         **roof(100000 * (12 + 4 + 4 + 12) + F * (36 + 36), tge, "hbm + float atomics on the face gradients"))
    nq = 20 * F
    q_d = torch.from_numpy(rng.uniform(-0.3, 0.3, (1, nq, 3)).astype(np.float32)).to(dev)
    tg = gpu_time(lambda: hip_ops.nn_index(q_d, gt_d), reps=3)
    subq = 200
    tc = cpu_time(lambda: O.nn_index(q_d[:, :subq].cpu().numpy(), gt[None])) * (nq / subq)
    tb = gpu_time(lambda: hip_ops.nn_index(q_d, gt_d, brute=True), reps=3)
    emit(op="nn_index", queries="uniform in the cube (far from the point cloud)", n_query=nq, n_point=100000, gpu_ms=round(tg * 1e3, 3),
         gpu_brute_ms=round(tb * 1e3, 3), cpu_ms=round(tc * 1e3, 1), cpu_kind="port (extrapolated from %d queries)" % subq, cpu_cores=1,
         pairs_per_s=round(nq * 1e5 / tg / 1e9, 2), unit="G nominal distance evals/s")
    # the training-time distribution: 20 samples per predicted boundary face (deftet.py:174-177), i.e. queries ON a
    # surface close to the ground-truth cloud
    from deftet_amd import surface_losses
    qs = surface_losses.sample_on_faces(face_d[None], 20).reshape(1, -1, 3).contiguous()
    tg2 = gpu_time(lambda: hip_ops.nn_index(qs, gt_d), reps=3)
    tb2 = gpu_time(lambda: hip_ops.nn_index(qs, gt_d, brute=True), reps=3)
    tge = gpu_time_events(lambda: hip_ops.nn_index(qs, gt_d))
    emit(op="nn_index", queries="20 samples per boundary face (training distribution)", n_query=int(qs.shape[1]), n_point=100000,
         gpu_ms=round(tg2 * 1e3, 3), gpu_brute_ms=round(tb2 * 1e3, 3), device_ms=round(tge * 1e3, 4),
         **roof(int(qs.shape[1]) * (12 + 8) + 100000 * 12, tge, "valu + launch count (cell sort of the cloud, then a ring search per query)"),
         pairs_per_s=round(qs.shape[1] * 1e5 / tg2 / 1e9, 2), unit="G nominal distance evals/s")

    # rasterizer, BASELINE configs[4]
    from tests.test_raster_gpu import pixel_grid, projected_grid
    from deftet_amd.render import deftet_sparse_render
    fz, fxy, ffe = projected_grid(40 if quick else 70)
    npx = 256 if quick else 512
    pix, rngs = pixel_grid(npx)
    t = [torch.from_numpy(x).to(dev) for x in (pix, rngs, fz, fxy, ffe)]
    t[3].requires_grad_(True)
    t[4].requires_grad_(True)
    tg = gpu_time(lambda: deftet_sparse_render(*t, knum=64), reps=3)
    feat, face_i = deftet_sparse_render(*t, knum=64)
    go = torch.rand_like(feat)
    tb = gpu_time(lambda: torch.autograd.grad(feat, (t[3], t[4]), go, retain_graph=True), reps=3)
    subp = 64
    tc = cpu_time(lambda: O.sparse_render_fwd(pix[:, :subp], rngs[:, :subp], fz, fxy, ffe, knum=64)) * (pix.shape[1] / subp)
    Fn, Pn = fxy.shape[1], pix.shape[1]
    emit(op="deftet_sparse_render", n_pixel=Pn, n_face=Fn, knum=64, fwd_ms=round(tg * 1e3, 2), bwd_ms=round(tb * 1e3, 2),
         cpu_fwd_ms=round(tc * 1e3, 0), cpu_kind="port (brute force, extrapolated from %d pixels)" % subp, cpu_cores=1,
         nominal_pixel_face_tests_per_s=round(Pn * Fn / tg / 1e9, 1), unit="G pixel-face tests/s (fwd)",
         hits=int((face_i >= 0).sum().item()), parity="unpinned (Kaolin not in the reference tree)")

    # ---- N2: vertex <-> tet gather (deftet.py:65-68) against torch's own gather / scatter-add on the same GPU
    Bv = 2 if quick else 8
    posv = torch.from_numpy(grids.jittered_positions(verts, res, Bv, 0.1).astype(np.float32)).to(dev)
    idx1 = tets_d.long()
    idxB = idx1[None].expand(Bv, -1, -1).contiguous()
    gt = torch.randn(Bv, T, 4, 3, device=dev)
    csr = hip_ops.tet_vertex_csr(idx1, n_point)
    t_csr = gpu_time(lambda: hip_ops.tet_vertex_csr(idx1, n_point))
    t_f = gpu_time(lambda: hip_ops.tet_gather(posv, idx1), reps=10)
    t_b = gpu_time(lambda: hip_ops.tet_gather_bwd(gt, csr, n_point), reps=10)

    def torch_fwd():
        return torch.gather(posv.unsqueeze(2).expand(-1, -1, 4, -1), 1, idxB.unsqueeze(-1).expand(-1, -1, -1, 3))

    def torch_bwd():
        out = torch.zeros(Bv, n_point, 3, device=dev)
        return out.scatter_add_(1, idxB.reshape(Bv, -1, 1).expand(-1, -1, 3), gt.reshape(Bv, -1, 3))

    tt_f, tt_b = gpu_time(torch_fwd, reps=10), gpu_time(torch_bwd, reps=10)
    emit(op="tet_gather", res=res, batch=Bv, n_tet=T, n_vertex=n_point, fwd_ms=round(t_f * 1e3, 4), bwd_ms=round(t_b * 1e3, 4),
         csr_build_ms=round(t_csr * 1e3, 3), torch_gather_ms=round(tt_f * 1e3, 4), torch_scatter_add_ms=round(tt_b * 1e3, 4),
         bwd_speedup_vs_torch=round(tt_b / t_b, 1), note="torch ops run on the same MI355X; bwd bytes = %.1f MB read" % (Bv * T * 48 / 1e6))

    # ---- N3: render-side rebuilds (prepare_for_wz.py) vs the vectorised numpy port (the reference's own
    # matchedgelist is O(E*T) Python: ~E*T*1e-8 s, hours at this size)
    t64 = tets.astype(np.int64)
    ptsn = np.random.default_rng(0).standard_normal((n_point, 3)).astype(np.float32)
    featn = np.random.default_rng(1).standard_normal((n_point, 8)).astype(np.float32)
    td, pd, fd = torch.from_numpy(t64).to(dev), torch.from_numpy(ptsn).to(dev), torch.from_numpy(featn).to(dev)
    tg = gpu_time(lambda: hip_ops.subdivide(td, pd, fd), reps=3)
    tc = cpu_time(lambda: O.generate_subdivision(t64, ptsn, featn))
    emit(op="generate_subdivision", res=res, n_tet=T, n_point=n_point, gpu_ms=round(tg * 1e3, 3), cpu_ms=round(tc * 1e3, 1),
         cpu_kind="port (vectorised numpy)", cpu_cores=1, speedup=round(tc / tg, 1))
    tg = gpu_time(lambda: hip_ops.point_adj_idx(n_point, td), reps=3)
    tc = cpu_time(lambda: O.generate_point_adj_idx(n_point, t64))
    emit(op="generate_point_adj_idx", res=res, n_tet=T, n_point=n_point, gpu_ms=round(tg * 1e3, 3), cpu_ms=round(tc * 1e3, 1),
         cpu_kind="port (vectorised numpy; the reference allocates a dense %d x %d float matrix = %.1f GB)" % (n_point, n_point, n_point * n_point * 4 / 1e9),
         cpu_cores=1, speedup=round(tc / tg, 1))

    # ---- N1: check_sign of all tet centroids against a closed surface (layers/DefTet/deftet.py:33-49)
    Bc = 2 if quick else 8
    posc = grids.jittered_positions(verts, res, Bc, 0.1).astype(np.float32)
    f3, t2, _, _, _ = O.tet_to_face(tets, n_point)
    occ = np.linalg.norm((verts - 0.5)[tets].mean(1), axis=1) < 0.3
    faces = f3[occ[t2].sum(1) == 1].astype(np.int64)
    cen = posc[:, tets].mean(2).astype(np.float32)
    vd, fdv, cd = torch.from_numpy(posc).to(dev), torch.from_numpy(faces).to(dev), torch.from_numpy(cen).to(dev)
    tg = gpu_time(lambda: hip_ops.check_sign(vd, fdv, cd, check=False), reps=5)
    tbr = gpu_time(lambda: hip_ops.check_sign(vd, fdv, cd, brute=True, check=False), reps=2, warm=1)
    sub = 2000
    tc = cpu_time(lambda: O.check_sign(posc[:1], faces, cen[:1, :sub])) * (cen.shape[1] / sub) * Bc
    emit(op="check_sign", res=res, batch=Bc, n_point=int(cen.shape[1]), n_face=int(faces.shape[0]), gpu_ms=round(tg * 1e3, 3),
         gpu_brute_ms=round(tbr * 1e3, 2), cpu_ms=round(tc * 1e3, 0), cpu_kind="port (OpenMP, extrapolated from %d points)" % sub,
         cpu_cores=os.cpu_count(), nominal_ray_face_tests_per_s=round(Bc * cen.shape[1] * faces.shape[0] / tg / 1e12, 2),
         unit="T ray-face tests/s", parity="unpinned (Kaolin not in the reference tree)")

    # ---- A11: fused per-tet energies (layers/DefTet/deftet.py:239-338) vs the same expressions as torch ops (fp32, same GPU)
    Be = 2 if quick else 8
    tete = torch.from_numpy(grids.gather_tets(grids.jittered_positions(verts, res, Be), tets)).to(dev).requires_grad_(True)
    rest = torch.from_numpy((verts - 0.5).astype(np.float32)).to(dev)[tets_d.long()] * 20
    inv = torch.inverse(torch.stack([rest[:, 1] - rest[:, 0], rest[:, 2] - rest[:, 0], rest[:, 3] - rest[:, 0]], 1))
    gsel = torch.ones(Be, 3, device=dev)

    def fused():
        out = hip_ops.tet_energies(tete, inv, 4, 4, 20.0)
        return torch.autograd.grad((out * gsel).sum(), tete)

    def torch_ref():
        A, Bv, C, Dd = (tete[:, :, i] for i in range(4))
        V = -(torch.cross(Bv - Dd, C - Dd, dim=-1) * (A - Dd)).sum(-1) / 6
        vv = ((V - V.mean(-1, keepdim=True)) ** 4).sum(-1)
        off = torch.stack([Bv * 20 - A * 20, C * 20 - A * 20, Dd * 20 - A * 20], 2)
        J = off @ inv[None]
        det = (J[..., 0, :] * torch.cross(J[..., 1, :], J[..., 2, :], dim=-1)).sum(-1)
        am = ((J ** 2).sum((-1, -2)) * (det ** 2 + 1e-10) ** (-1.0 / 3.0) * (det >= 0)).mean(-1)
        el = sum((((p - q) * 20.0) ** 4).sum(-1).sum(-1) for p, q in ((A, Dd), (Bv, Dd), (C, Dd), (A, Bv), (A, C), (Bv, C))) / (6 * T)
        return torch.autograd.grad((torch.stack([vv, am, el], -1) * gsel).sum(), tete)

    tf_, tt_ = gpu_time(fused, reps=10), gpu_time(torch_ref, reps=5)
    tfe = gpu_time_events(fused, reps=30)
    # forward reads the 48-byte record once (+ 4 B saved volume written and read back); backward reads it and writes 48 B
    emit(op="tet_energies fwd+bwd", res=res, batch=Be, n_tet=T, gpu_ms=round(tf_ * 1e3, 3), device_ms=round(tfe * 1e3, 4),
         torch_same_gpu_ms=round(tt_ * 1e3, 3), speedup_vs_torch=round(tt_ / tf_, 1),
         **roof(Be * T * (48 + 48 + 48) + T * 36 * 2, tfe, "hbm"))

    # ---- A1 forward: the binned path against the brute-force HIP formulation (the algorithmic equivalent of the
    # reference kernel: every query meets every tet in index order), BASELINE configs[2]
    Bp, Qp = (2, 20000) if quick else (8, 100000)
    tetp = torch.from_numpy(grids.gather_tets(grids.jittered_positions(verts, res, Bp), tets)).to(dev)
    ptsp = torch.from_numpy(grids.random_queries(Bp, Qp)).to(dev)
    t_bin = gpu_time(lambda: hip_ops.point_in_tet(tetp, ptsp, want_bary=True), reps=10)
    t_bru = gpu_time(lambda: hip_ops.point_in_tet(tetp, ptsp, want_bary=True, algo=1), reps=2, warm=1)
    emit(op="point_in_tet forward (index + weights)", res=res, batch=Bp, n_tet=T, n_query=Qp, binned_ms=round(t_bin * 1e3, 3),
         brute_hip_ms=round(t_bru * 1e3, 2), speedup=round(t_bru / t_bin, 1),
         brute_pairs_per_s=round(Bp * T * Qp / t_bru / 1e12, 2), unit="T tet-point tests/s (brute)")


if __name__ == "__main__":
    main()
