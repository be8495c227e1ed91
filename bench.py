#!/usr/bin/env python
"""bench.py — headline benchmark of the DefTet per-tetrahedron hot path on MI355X.

Metric (BASELINE.json): M tet-point tests/s (fwd+bwd) at res=70, 100k queries.
One "step" = one pass of the hot path over one batch of B=8 synthetic shapes per GPU:
    fwd : point-in-tet index (A1) + barycentric weights of the hit tet + paste_occ gather
    bwd : dL/dtet scatter (A1b) + dL/dpred scatter (paste_occ backward) + per-shape loss scalars
`value` counts NOMINAL tet-point pairs B*T*Q per step (what the reference's brute-force kernel
enumerates), inputs resident in HBM, acceleration-structure build included.  The steps rotate over
N_SETS distinct input sets (different deformed grids, queries and gradients) so that no step finds
its inputs in the 256 MiB Infinity Cache left by the previous one.

    python bench.py [--gpus N --steps K --warmup W] [--config {1,2,3,4}]

--gpus N > 1 without a torch.distributed environment re-launches itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N` (one rank per GPU, RCCL);
when the driver has already launched it that way (WORLD_SIZE set) it just runs as a rank.
--config picks the BASELINE.json configuration that is timed as the main line (default 2 =
configs[2], the one the metric is quoted on); at N=1 the other configurations are measured after the
timed region and reported under "other_configs" in the same JSON line.
"""
from __future__ import annotations

import argparse
import ctypes
import gc
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: 8 TB/s HBM3E spec
N_SETS = 3                                    # distinct input sets the steps rotate over

# BASELINE.json configs[i] -> workload.  configs[3] is "batch=64 sharded 8 shapes/GPU across 8 GPUs":
# 8 shapes per GPU, so `--gpus 8 --config 3` IS that configuration and `--gpus 1 --config 3` one GPU's share.
CONFIGS = {
    1: dict(kind="pit", res=40, n_query=50_000, batch=8, sets=3,
            name="BASELINE configs[1]: res=40 Kuhn tet grid, 50k queries, batch=8"),
    2: dict(kind="pit", res=70, n_query=100_000, batch=8, sets=N_SETS,
            name="BASELINE configs[2]: res=70 Kuhn tet grid (paper config), 100k queries, batch=8"),
    3: dict(kind="pit", res=100, n_query=200_000, batch=8, sets=2,
            name="BASELINE configs[3]: res=100 Kuhn tet grid, 200k queries, 8 shapes per GPU (batch=64 over 8 GPUs)"),
    4: dict(kind="raster", res=70, npx=512, knum=64, batch=1, sets=1,
            name="BASELINE configs[4]: tet rasterizer, 512x512 rays x k=64 over the unique faces of the res=70 grid"),
    # not a BASELINE config: the geometry side of one TRAINING step (layers/DefTet/deftet.py:51-130) — forward_surface_align
    # incl. the per-shape surface terms A8/A9/A10 + the occupancy query, forward and backward down to the vertices
    5: dict(kind="geometry", res=70, n_query=100_000, n_gt=100_000, batch=8, sets=1,
            name="geometry step incl. surface terms: res=70 grid, 100k queries, 100k ground-truth surface points, batch=8"),
}

ALGO = int(os.environ.get("DEFTET_BENCH_ALGO", "0"))      # A/B switch for the traversal kernel (0 = shipped default)
PIPELINE = os.environ.get("DEFTET_BENCH_PIPELINE", "0") not in ("", "0")   # cross-step overlap of the query sort: off since round 3 (measured: no gain)


def dominant_kernel(algo, n_tet, n_query):
    from deftet_amd import hip_ops
    # (DEFTET_BENCH_KERNEL: name of the traversal kernel when DEFTET_HIP_LIB points at a probe build with other kernels)
    return (os.environ.get("DEFTET_BENCH_KERNEL") or hip_ops.pit_kernel_name(algo, n_tet, n_query)).encode()


class PitWorkload:
    """point-in-tet fwd+bwd (A1 + A1b + paste_occ) on `sets` rotating input sets."""

    def __init__(self, cfg, rank, device, world, gather=None, pipeline=PIPELINE, algo=ALGO):
        from deftet_amd import grids
        self.cfg, self.world, self.gather, self.pipeline, self.algo = cfg, world, gather, pipeline, algo
        self.device = torch.device(device)
        res, Q, B = cfg["res"], cfg["n_query"], cfg["batch"]
        if cfg.get("mesh") == "cube40":                   # the shipped QuarTet grid (probe use: a non-Kuhn tet order)
            g40 = np.load(os.path.join(ROOT, "tests", "golden", "cube40_grid.npz"))
            verts, tets = g40["verts"], g40["tets"]
        else:
            verts, tets = grids.kuhn_grid(res)
        if cfg.get("order") == "xfast":                   # probe use: cubes enumerated with x fastest instead of z fastest
            n = res // 2
            idx = np.arange(n * n * n * 6).reshape(n, n, n, 6)       # [ix, iy, iz, k] -> current position
            tets = tets[idx.transpose(2, 1, 0, 3).reshape(-1)]       # new position ((iz*n + iy)*n + ix)*6 + k
        if cfg.get("mesh") == "shuffled":                 # probe use: the same grid, its tet list in random order
            rng7 = np.random.default_rng(7)
            frac = float(cfg.get("shuffle_frac", 1.0))      # < 1: only that fraction of the positions trade places (probe: the order rule's crossover)
            if frac >= 1.0:
                self.shuffle_perm = rng7.permutation(tets.shape[0])
            else:
                sel = np.sort(rng7.choice(tets.shape[0], int(frac * tets.shape[0]), replace=False))
                self.shuffle_perm = np.arange(tets.shape[0])
                self.shuffle_perm[sel] = sel[rng7.permutation(sel.size)]
            tets = tets[self.shuffle_perm]
        # traversal order handed to the operator: "auto" (what the autograd ops pass: decided once per grid), "native" (none),
        # "sorted" (the computed column order, unconditionally) — DEFTET_BENCH_TET_ORDER / cfg["tet_order"]; never changes a result
        self.order_mode = cfg.get("tet_order") or os.environ.get("DEFTET_BENCH_TET_ORDER", "auto")
        # query grid box: "track" (what the autograd ops pass: the box the previous step's queries measured, one launch fewer;
        # the steps rotate over input sets with DIFFERENT queries) or "measure" (every step measures its own) — DEFTET_BENCH_QUERY_BOX
        qb = cfg.get("query_box") or os.environ.get("DEFTET_BENCH_QUERY_BOX", "track")
        self.query_box = "track" if qb == "track" else None
        self.sets, self.host = [], None
        # N > 1: the sets are generated ON THE GPU (same distributions, torch generators seeded per rank and set): eight
        # ranks x several sets of res-100 numpy jitter + gathers on one host are minutes of CPU before the first step.  N = 1
        # keeps the numpy generators of deftet_amd.grids (bit-identical to the parity tests' inputs, and the CPU baseline needs
        # a host copy).
        on_gpu = world > 1 and cfg.get("generate", "auto") != "host"
        verts_d = torch.from_numpy(verts).to(device) if on_gpu else None
        tets_d = torch.from_numpy(tets.astype(np.int64)).to(device) if on_gpu else None
        for s in range(cfg["sets"]):
            base = rank * B + s * 100_000                 # distinct seeds per rank and per set
            if on_gpu:
                g = torch.Generator(device=device).manual_seed(1000 + base)
                h = 2.0 / res
                interior = ((verts_d > 0) & (verts_d < 1)).to(torch.float64)
                d = (torch.rand((B,) + tuple(verts_d.shape), device=device, dtype=torch.float64, generator=g) * 0.2 - 0.1) * h
                pos = (verts_d[None] - 0.5 + d * interior[None]).to(torch.float32)
                T = tets_d.shape[0]
                dset = dict(tet=pos[:, tets_d, :].contiguous(),
                            pts=(1.05 * (torch.rand(B, Q, 3, device=device, dtype=torch.float64, generator=g) - 0.5)).to(torch.float32),
                            gw=torch.randn(B, Q, 4, device=device, generator=g), pred=torch.rand(B, T, device=device, generator=g),
                            gout=torch.randn(B, Q, device=device, generator=g))
                del pos, d
                self.sets.append(dset)
                continue
            pos = grids.jittered_positions(verts, res, B, 0.1, seed0=1000 + base)
            tet = grids.gather_tets(pos, tets)
            pts = grids.random_queries(B, Q, seed0=2000 + base)
            T = tet.shape[1]
            gw = np.stack([np.random.default_rng(4000 + base + b).standard_normal((Q, 4)).astype(np.float32) for b in range(B)])
            pred = np.stack([np.random.default_rng(5000 + base + b).random(T).astype(np.float32) for b in range(B)])
            gout = np.stack([np.random.default_rng(6000 + base + b).standard_normal(Q).astype(np.float32) for b in range(B)])
            if s == 0:
                self.host = dict(tet=tet[:1].copy(), pts=pts[:1].copy())
            self.sets.append({k: torch.from_numpy(v).to(device) for k, v in dict(tet=tet, pts=pts, gw=gw, pred=pred, gout=gout).items()})
        self.generated_on = "gpu" if on_gpu else "host"
        self.B, self.T, self.Q = B, self.sets[0]["tet"].shape[1], Q
        self.pairs_per_step = float(B) * self.T * Q
        self.unit = "M tet-point tests/s"
        self.dominant = dominant_kernel(algo, self.T, Q)
        # SURVEY.md 8(d), A1 fwd: B*(48*T + 12*Q + 4*Q) algorithmic bytes per call — what the traversal kernel must
        # touch once (48-byte tet records, 16-byte sorted queries); its per-tet hit records and the result atomics are
        # overhead of THIS design and are not counted
        self.dominant_bytes = B * (48.0 * self.T + 16.0 * Q)
        # whole step: fwd with weights B*(48T+16Q+16Q), bwd B*(32Q + 48*hits + 48T), hits ~ 0.864*Q
        self.step_bytes = B * (48.0 * self.T + 32.0 * Q) + B * (32.0 * Q + 48.0 * 0.864 * Q + 48.0 * self.T)
        self._side = None
        self._pq = {}
        self.last = None
        from deftet_amd import hip_ops
        if self.order_mode == "sorted":
            self.order = hip_ops.tet_spatial_order(self.sets[0]["tet"][0])
        elif self.order_mode == "auto":
            self.order = hip_ops.auto_tet_order(self.sets[0]["tet"], self.sets[0]["pts"], algo)   # resolved here: the steps pass a tensor or None
        elif self.order_mode == "identity":               # probe: the ordered kernel instance on the caller's own order
            self.order = torch.arange(self.T, device=device, dtype=torch.int32)
        elif self.order_mode == "ideal":                  # probe (mesh "shuffled"): the permutation that undoes the shuffle
            self.order = torch.from_numpy(np.argsort(self.shuffle_perm).astype(np.int32)).to(device)
        else:
            self.order = None

    def side_stream(self):
        if self._side is None:
            # (stream priorities — the step on a high-priority stream, the overlapped sort on a normal one — were measured:
            # no difference, 0.243-0.245 ms/step either way; gfx950 exposes two levels only.  A side stream confined to
            # 16 / 32 / 64 / 128 compute units with hipExtStreamCreateWithCUMask: 0.63 / 0.52 / 0.46 / 0.43 ms/step.
            # Reducing the per-shape loss scalars on the side stream while the backward runs: 0.246 vs 0.245.)
            self._side = torch.cuda.Stream()
        return self._side

    def step(self, i):
        """fwd: index + weights + fused paste_occ gather (+ per-tet hit records); bwd: dL/dtet and dL/dpred from one
        per-tet pass over those records (no atomics); then the per-shape loss scalars (all-gathered when world > 1).
        Same calls as the PointInTetOcc autograd op makes."""
        from deftet_amd import hip_ops
        d = self.sets[i % len(self.sets)]
        if self.pipeline:
            # software pipelining across steps: the query side of the operator (bounding box + counting sort, small
            # latency-bound kernels that need only the points) of step i+1 is enqueued on a second stream as soon as
            # step i's traversal has been launched; every step still does all of its work (K sorts in K timed steps)
            pq = self._pq.pop(i, None)
            if pq is None:
                pq = hip_ops.prepare_queries(d["pts"], self.T, algo=self.algo, query_box=self.query_box)
            cond, w, occ, hits = hip_ops.point_in_tet(d["tet"], d["pts"], want_bary=True, pred_bxt=d["pred"], want_hits=True,
                                                      algo=self.algo, prepared=pq, order=self.order)
            nxt = self.sets[(i + 1) % len(self.sets)]
            # (making the side stream wait for this step's forward, so that the sort overlaps the HBM-bound backward instead
            # of the traversal, was measured: 0.257-0.259 vs 0.247-0.248 ms/step)
            with torch.cuda.stream(self.side_stream()):
                self._pq = {i + 1: hip_ops.prepare_queries(nxt["pts"], self.T, algo=self.algo, query_box=self.query_box)}
        else:
            cond, w, occ, hits = hip_ops.point_in_tet(d["tet"], d["pts"], want_bary=True, pred_bxt=d["pred"], want_hits=True,
                                                      algo=self.algo, order=self.order, query_box=self.query_box)
        g_tet, _, g_pred = hip_ops.point_in_tet_bwd(d["tet"], d["pts"], cond, d["gw"], grad_occ=d["gout"], hits=hits)
        loss = hip_ops.rowdot(w, d["gw"], occ, d["gout"])         # [B] per-shape loss scalars
        if self.world > 1:
            # the only collective (RCCL over xGMI): enqueued asynchronously, consumed one step later
            loss = self.gather.submit(loss.cpu() if torch.distributed.get_backend() == "gloo" else loss)
        self.last = (cond, w, g_tet, g_pred, loss)

    def drain(self):
        self._pq = {}
        if self.world > 1 and self.gather is not None:
            self.gather.flush()                           # the last step's all-gather belongs to the timed region

    def _tracker_counts(self):
        from deftet_amd import hip_ops
        st = hip_ops.query_box_trackers().get(hip_ops.query_box_key(self.device, self.B, self.Q))
        return "no tracked call yet" if st is None else "%d tracked, %d measured, %d fall-backs to measuring" % (st["tracked"], st["measured"], st["backoffs"])

    def describe(self):
        return {"workload": "%s: T=%d tets, %d uniform queries, %d shapes per GPU, point-in-tet index + weights + paste_occ, "
                            "fwd+bwd, grid build included, %d rotating input sets" % (self.cfg["name"], self.T, self.Q, self.B, len(self.sets)),
                "res": self.cfg["res"], "n_tet": self.T, "n_query": self.Q, "batch_per_gpu": self.B, "input_sets": len(self.sets),
                "sharding": "shapes sharded by rank; all-gather of %d loss scalars" % (self.world * self.B),
                "inputs_generated_on": self.generated_on,
                "tet_order": "%s -> %s" % (self.order_mode, "the caller's numbering" if self.order is None else "computed column order"),
                "query_box": ("tracked: the grid of a step spans the box the previous step's (different) queries measured; queries outside it "
                              "take the exact side path and are counted; calls so far: %s" % self._tracker_counts()
                              if self.query_box else "measured by every step"),
                "pipelining": ("query sort of step i+1 enqueued on a second stream during step i" if self.pipeline else "none")}


class RasterWorkload:
    """deftet_sparse_render fwd+bwd at BASELINE configs[4] (parity unpinned w.r.t. Kaolin)."""

    def __init__(self, cfg, rank, device, world, gather=None):
        from deftet_amd import grids, hip_ops
        self.cfg, self.world = cfg, world
        verts, tets = grids.kuhn_grid(cfg["res"])
        f3 = hip_ops.tet_to_face(tets, verts.shape[0], device, with_boundary=True)[0].cpu().numpy()
        fz, fxy, ff = grids.project_faces(verts, f3, seed=rank)
        pix, rngs = grids.pixel_grid(cfg["npx"])
        self.t = [torch.from_numpy(x).to(device) for x in (pix, rngs, fz, fxy, ff)]
        self.t[3].requires_grad_(True)
        self.t[4].requires_grad_(True)
        self.P, self.F, self.k = pix.shape[1], fxy.shape[1], cfg["knum"]
        self.go = torch.rand(1, self.P, self.k, 4, device=device, generator=torch.Generator(device=device).manual_seed(rank))
        self.pairs_per_step = float(self.P) * self.F
        self.unit = "M ray-face tests/s"
        self.dominant = b"k_pix_raster"
        self.valu_files = ("profiles/r06_pmc_raster.json",)       # (counters of THIS round's kernel only)
        # SURVEY 8(d) A12 fwd: F*(12+24+48) + P*(8+8) + P*k*(16+4)
        self.dominant_bytes = self.F * 84.0 + self.P * 16.0 + self.P * self.k * 20.0
        self.step_bytes = 2.0 * self.dominant_bytes
        self.last = None
        # saturation policy (include/deftet_hip.h): 0 = NEAREST (the library default), 1 = FIRST; DEFTET_BENCH_RASTER_POLICY
        self.policy = int(os.environ.get("DEFTET_BENCH_RASTER_POLICY", "0"))

    def saturated(self):
        from deftet_amd.render.deftet_sparse_render import saturated_pixels
        return saturated_pixels(*[x.detach() for x in self.t], knum=self.k)

    def step(self, i):
        from deftet_amd.render import deftet_sparse_render
        feat, face = deftet_sparse_render(*self.t, knum=self.k, policy=self.policy)
        gxy, gff = torch.autograd.grad(feat, (self.t[3], self.t[4]), self.go)
        self.last = (feat, face, gxy, gff)

    def drain(self):
        pass

    def describe(self):
        return {"workload": "%s: %d rays x %d faces, k=%d (%s saturation policy), fwd+bwd, tile binning included; PARITY UNPINNED "
                            "(Kaolin absent)" % (self.cfg["name"], self.P, self.F, self.k, ["nearest-k", "first-k"][self.policy]),
                "n_ray": self.P, "n_face": self.F, "knum": self.k}


class GeometryWorkload:
    """DefTet.forward_surface_align (training branch) + point-in-tet occupancy, fwd + bwd (tools/step_demo.py --surface)."""

    def __init__(self, cfg, rank, device, world, gather=None):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import step_demo
        from deftet_amd import surface_losses
        from deftet_amd.layers.DefTet.deftet import DefTet
        self.cfg, self.demo = cfg, step_demo
        B = cfg["batch"]
        pos0, idx, f3, t2, gt_verts, gt_faces, pts, inv_v = step_demo.build_case(cfg["res"], B, cfg["n_query"], device)
        per_face = max(1, cfg["n_gt"] // max(1, gt_faces.shape[0]))
        tri = gt_verts[gt_faces.long()][None].expand(B, -1, -1, -1)
        self.gt = surface_losses.sample_on_faces(tri, per_face, torch.Generator(device=device).manual_seed(5)).reshape(B, -1, 3).contiguous()
        self.pos = pos0.clone().requires_grad_(True)
        self.pred = torch.rand(B, idx.shape[0], device=device, requires_grad=True)
        self.m = DefTet(device=device)
        self.args = (idx[None].expand(B, -1, -1).contiguous(), f3, t2, gt_verts, gt_faces, pts, inv_v)
        self.B, self.T, self.Q = B, idx.shape[0], cfg["n_query"]
        self.pairs_per_step = float(B)
        self.unit = "shapes/s"
        self.dominant = b"k_tri_query_coop"
        self.valu_files = ("profiles/r06_pmc_geometry.json", "profiles/r05_pmc_geometry.json")
        self.dominant_bytes = 0.0
        self.step_bytes = 0.0
        self.last = None

    def step(self, i):
        self.pos.grad = None
        self.pred.grad = None
        self.last = self.demo.run_full_step(self.m, self.pos, *self.args, self.pred, self.gt)

    def drain(self):
        pass

    def describe(self):
        return {"workload": "%s: T=%d tets, %d GT points and %d queries per shape, DefTet.forward_surface_align (gather, check_sign, boundary "
                            "faces, energies, ragged A8/A9/A10 surface terms) + occupancy query, fwd + bwd to the vertices; the largest "
                            "kernel (A9 k_tri_query_coop) is VALU-bound: its roofline is the issue fraction" % (
                                self.cfg["name"], self.T, self.gt.shape[1], self.Q)}


def make_workload(cfg_id, rank, device, world, gather=None, **kw):
    cfg = CONFIGS[cfg_id]
    return {"pit": PitWorkload, "raster": RasterWorkload, "geometry": GeometryWorkload}[cfg["kind"]](cfg, rank, device, world, gather, **kw)


def timed(wl, lib, steps, warmup, world, barrier=True, step_events=False):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides.  Returns the wall
    time, the per-step HIP-event times (ms; empty unless step_events) and the dominant kernel's (total ms, launches)
    from the library's own events on the launch stream."""
    # Everything the timed region would otherwise do lazily is done BEFORE the warm-up: the step events exist and have
    # been recorded once, the cyclic garbage collector has run and is off (one timed region of 20 steps was once 45 ms
    # long with a median step of 0.229 ms, profiles/README.md: a single host-side stall), and the library's launch events
    # exist (selected for the warm-up already, those samples dropped below).  Between the last warm-up step and the first
    # timed one there is then nothing but the barrier and the synchronize the contract asks for: rounds 1-3 created the
    # events and collected garbage THERE, the GPU sat idle for milliseconds and the first timed step ran 0.15 ms longer
    # than the others (7 us per step of a 20-step mean).
    # step_events: an event after every step gives the median / maximum, but every event record is a barrier packet in
    # the queue (3-4 us per step at configs[2]); the headline region is timed without them, a second region with.
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if step_events else []
    for e in evs:
        e.record()
    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    lib.deftet_profile_select(wl.dominant)
    for i in range(warmup):
        wl.step(i)
    wl.drain()
    if world > 1 and barrier:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    tot_ms, cnt = ctypes.c_double(0), ctypes.c_longlong(0)
    lib.deftet_profile_read(ctypes.byref(tot_ms), ctypes.byref(cnt))
    lib.deftet_profile_select(wl.dominant)
    t0 = time.perf_counter()
    for i in range(steps):
        if step_events:
            evs[i].record()
        wl.step(warmup + i)
    if step_events:
        evs[steps].record()
    wl.drain()
    torch.cuda.synchronize()
    if world > 1 and barrier:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()
    tot_ms, cnt = ctypes.c_double(0), ctypes.c_longlong(0)
    lib.deftet_profile_read(ctypes.byref(tot_ms), ctypes.byref(cnt))
    lib.deftet_profile_select(b"")
    per_step = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)] if step_events else []
    return elapsed, per_step, tot_ms.value, cnt.value


def measure_bandwidth(device):
    """What this box's HBM delivers to a plain streaming kernel (deftet_bandwidth_probe: float4, nontemporal, 32 KB per
    workgroup): a 1 GiB device-to-device copy (read + write bytes) and a read-only pass, HIP-event timed.  Quoted beside
    the 8 TB/s datasheet figure; MI355X_MICROARCH.md measures 6.29 TB/s for a float4 copy on this part."""
    from deftet_amd import _lib
    lib = _lib.load()
    n = 1 << 30
    src = torch.empty(n // 4, device=device, dtype=torch.float32).uniform_()
    dst = torch.empty_like(src)
    st = _lib.current_stream(device)
    out = {}
    for name, mode, nbytes in (("copy", 0, 2.0 * n), ("read", 1, 1.0 * n)):
        def fn():
            _lib.check(lib.deftet_bandwidth_probe(_lib.ptr(src), _lib.ptr(dst), n, mode, None, st), "deftet_bandwidth_probe")
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        out[name] = nbytes * reps / (a.elapsed_time(b) * 1e-3) / 1e9
    del src, dst
    return out


def brute_force_comparator(wl, step_ms):
    """north_star's "10x the brute-force path": the library's own brute-force kernel (DEFTET_PIT_BRUTE: every query meets every
    tet in index order, the algorithmic equivalent of check_condition_tet_for.cu:124-189 written for CDNA4) on input set 0 of
    this workload — the whole batch, as the step has it — outside the timed region, HIP-event timed.  The ratio compares its
    FORWARD ALONE with the whole binned forward + backward step, i.e. it is a lower bound of the speed-up."""
    from deftet_amd import hip_ops
    d = wl.sets[0]
    hip_ops.point_in_tet(d["tet"][:1].contiguous(), d["pts"][:1].contiguous(), algo=hip_ops.PIT_BRUTE)      # (loads the kernel)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ref = hip_ops.point_in_tet(d["tet"], d["pts"], algo=hip_ops.PIT_BRUTE)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    same = bool(torch.equal(ref, hip_ops.point_in_tet(d["tet"], d["pts"], algo=wl.algo, order=wl.order)))
    return {"kernel": "k_brute", "ms_fwd_batch": round(ms, 2), "M_tests_per_s_fwd": round(wl.pairs_per_step / (ms * 1e-3) / 1e6, 1),
            "speedup_of_binned_fwd_bwd_step_over_brute_fwd": round(ms / step_ms, 1), "same_result_as_binned": same,
            "how": "DEFTET_PIT_BRUTE forward on input set 0 (all %d shapes) after the timed region, one launch sequence, HIP events; "
                   "compared with the WHOLE binned fwd+bwd step" % wl.B}


def graph_replay(wl, steps, warmup=3):
    """The same K steps replayed from hipGraphs (one captured step per input set; torch.cuda.CUDAGraph is a hipGraph on
    ROCm): what the launch path costs is then out of the figure.  Reported beside the headline, never as `value`.
    Returns (ms_per_step, None) or (None, reason)."""
    if wl.world > 1 or wl.pipeline:
        return None, "single-GPU, unpipelined steps only"
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for k in range(len(wl.sets)):                # library workspaces reach their final size outside the capture
                wl.step(k)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graphs, keep = [], []
        for k in range(len(wl.sets)):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                wl.step(k)
            graphs.append(g)
            keep.append(wl.last)                         # the graph's output buffers
        for i in range(warmup):
            graphs[i % len(graphs)].replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            graphs[i % len(graphs)].replay()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        del graphs, keep
        return round(ms, 4), None
    except Exception as e:                               # a capture the runtime refuses is reported, not fatal
        torch.cuda.synchronize()
        return None, "%s: %s" % (type(e).__name__, str(e)[:200])


def cpu_baseline(wl):
    """Oracle (CPU restatement, kind="port") on a bounded sample of the same workload."""
    from oracle import oracle as O
    tet, pts = wl.host["tet"], wl.host["pts"]
    ncpu = os.cpu_count() or 1
    q1 = 1500
    t0 = time.perf_counter()
    O.point_in_tet(tet, pts[:1, :q1])
    t1 = time.perf_counter() - t0
    qn = min(wl.Q, max(2000, 1200 * ncpu))
    t0 = time.perf_counter()
    cond, nthreads = O.point_in_tet(tet, pts[:1, :qn], omp=True, return_executed=True)
    tn = time.perf_counter() - t0
    T = tet.shape[1]
    # the backward half of the metric on the same sample: weights (oracle_bary_f32) + dL/dtet (oracle_bary_bwd_f32, a plain
    # scatter-add), single-threaded as written — next to the all-pairs scan they are noise, which is the point of quoting them
    gw = np.random.default_rng(4000).standard_normal((1, qn, 4)).astype(np.float32)
    t0 = time.perf_counter()
    O.bary(tet, pts[:1, :qn], cond)
    O.bary_bwd(tet, pts[:1, :qn], cond, gw)
    tb = time.perf_counter() - t0
    model = "?"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": round(T * qn / tn / 1e6, 2), "unit": "M tet-point tests/s (fwd only)", "cores": int(nthreads),
        "kind": "port", "covers": "fwd",
        "covers_note": "the reference's own autograd backward for this op returns (None, None) "
                       "(check_condition_tetrahedron_base/utils.py:55-58); value_fwd_bwd adds this build's A1b backward on the CPU",
        "value_fwd_bwd": round(T * qn / (tn + tb) / 1e6, 2), "bwd_seconds_on_sample": round(tb, 4), "fwd_seconds_on_sample": round(tn, 3),
        "cpu_model": model, "nproc": ncpu,
        "sample": "oracle/deftet_oracle.c brute-force scan, 1 shape res=%d (T=%d), first %d queries, OpenMP over "
                  "queries; single-core on %d queries: %.2f M/s" % (wl.cfg["res"], T, qn, q1, T * q1 / t1 / 1e6),
        "value_1core": round(T * q1 / t1 / 1e6, 2),
    }


def traffic_record(kernel):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, corrected as
    MI355X_MICROARCH.md prescribes).  Counters cannot be read from inside the timed process, so this is the figure
    tools/pmc_traffic.py collected for the same command.  It is only quoted while the kernel source it was measured on
    (sha1 of deftet_amd/csrc/point_in_tet.hip, stored in the file) is still the one in the tree; otherwise null."""
    import hashlib
    for rel in ("profiles/r06_pmc_traffic.json", "profiles/r05_pmc_traffic.json", "profiles/r04_pmc_traffic.json", "profiles/r03_pmc_traffic.json"):
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        try:
            rec = json.load(open(path))
        except Exception:
            continue
        src = os.path.join(ROOT, "deftet_amd", "csrc", "point_in_tet.hip")
        now = hashlib.sha1(open(src, "rb").read()).hexdigest() if os.path.exists(src) else None
        v = rec.get("%s_hbm_bytes_per_launch" % kernel)
        if v is None:
            continue
        if rec.get("kernel_source_sha1") != now:
            return None, "%s (STALE: collected on another version of point_in_tet.hip, commit %s)" % (rel, rec.get("commit")), rec.get("commit")
        return v, rel, rec.get("commit")
    return None, None, None


VALU_PEAK_GINST = 256 * 4 * 2.4 / 4.0            # G wave-instructions/s: 1,024 SIMDs, one 64-lane VALU instruction per 4 cycles at 2.4 GHz


def valu_record(kernel, files):
    """SQ_INSTS_VALU per launch of `kernel` from a tools/pmc_run.sh summary (counters cannot be read from inside the timed
    process: the same rule as traffic_record — quoted with its source file, null when there is none)."""
    for rel in files:
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        try:
            rec = json.load(open(path))
        except Exception:
            continue
        for name, c in rec.items():
            if isinstance(c, dict) and kernel in name and "SQ_INSTS_VALU" in c:
                return c["SQ_INSTS_VALU"], rel
    return None, None


def summarize(wl, elapsed, per_step, kern_ms_tot, kern_cnt, steps, world, peak_measured=None):
    kern_ms = kern_ms_tot / max(kern_cnt, 1)
    achieved = wl.dominant_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    step_gbs = wl.step_bytes / (elapsed / steps) / 1e9
    kernel = wl.dominant.decode()
    traffic, src, tcommit = traffic_record(kernel)
    roof = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": src, "traffic_commit": tcommit,
            "algorithmic_bytes_per_launch": wl.dominant_bytes, "avg_launch_ms": round(kern_ms, 5), "launches_timed": int(kern_cnt),
            "whole_step": {"algorithmic_bytes": wl.step_bytes, "achieved": round(step_gbs, 1), "frac": round(step_gbs / HBM_PEAK_GBS, 4)}}
    if getattr(wl, "valu_files", None):
        # a VALU-bound kernel gets a VALU figure: wave-instructions per launch (rocprofv3 --pmc SQ_INSTS_VALU) over the launch's
        # duration measured HERE, against the chip's issue rate — next to the HBM fraction (rasterizer) or instead of it (A9 query)
        insts, src = valu_record(kernel, wl.valu_files)
        if insts and kern_ms > 0:
            ach = insts / (kern_ms * 1e-3) / 1e9
            roof["valu"] = {"bound": "valu", "achieved": round(ach, 1), "peak": round(VALU_PEAK_GINST, 1), "unit": "G wave-instructions/s",
                            "frac": round(ach / VALU_PEAK_GINST, 4), "insts_per_launch": insts, "insts_source": src}
            if wl.dominant_bytes <= 0:
                roof.update({"bound": "valu", "achieved": roof["valu"]["achieved"], "peak": roof["valu"]["peak"], "unit": roof["valu"]["unit"],
                             "frac": roof["valu"]["frac"]})
    if peak_measured:
        roof["peak_measured"] = {"copy_GBs": round(peak_measured["copy"], 1), "read_GBs": round(peak_measured["read"], 1),
                                 "how": "1 GiB float4 streaming kernel of the library (copy: read+write bytes; read-only pass) on this GPU, HIP events"}
        roof["frac_of_measured_read"] = round(achieved / peak_measured["read"], 4)
    return {
        "value": round(world * wl.pairs_per_step * steps / elapsed / (1e6 if wl.unit.startswith("M ") else 1.0), 1), "unit": wl.unit,
        "ms_per_step": round(elapsed / steps * 1e3, 4),
        "ms_per_step_median": round(statistics.median(per_step), 4) if per_step else None,
        "ms_per_step_max": round(max(per_step), 4) if per_step else None,
        "roofline": roof,
    }


def pin_to_gpu_numa_node(dev_index):
    """Best effort: restrict this rank's CPU threads to the cores of its GPU's NUMA node (eight ranks that all start on node 0
    share its memory controllers while they build their inputs and launch).  Returns a short description for the line."""
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return {"pci": bdf, "numa_node": None}
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
        return {"pci": bdf, "numa_node": node, "cpus": len(cpus)}
    except Exception as e:                                # no sysfs entry, no permission, an older torch: not fatal
        return {"pci": None, "numa_node": None, "note": "%s: %s" % (type(e).__name__, str(e)[:80])}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` started without a torch.distributed environment: become the launcher."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-brute-force", action="store_true", help="skip the brute-force HIP comparator (one shape, ~50 ms at configs[2])")
    ap.add_argument("--no-unpipelined", action="store_true", help="skip the extra un-overlapped timing loop (profiling runs: every traversal launch in the kernel table is then a timed-region launch)")
    ap.add_argument("--no-bandwidth-probe", action="store_true", help="skip the 1 GiB copy/read probe (profiling runs: keeps its launches out of the kernel table)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # DEFTET_BENCH_TEST_SHARED_GPU=1 is a TEST hook for single-GPU boxes: every rank uses cuda:0 and the
    # (CPU-staged) collectives run over gloo, so the whole multi-process control flow can be exercised
    # where only one GPU exists.  Real runs use one GPU per rank and RCCL ("nccl" backend).
    shared = os.environ.get("DEFTET_BENCH_TEST_SHARED_GPU", "0") not in ("", "0")
    dev_index = 0 if shared else local_rank
    if dev_index >= torch.cuda.device_count():
        raise SystemExit("rank %d wants cuda:%d but only %d device(s) are visible" % (rank, dev_index, torch.cuda.device_count()))
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    backend, placement, identities = None, None, None
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        placement = pin_to_gpu_numa_node(dev_index)
        limit = datetime.timedelta(seconds=int(os.environ.get("DEFTET_BENCH_INIT_TIMEOUT", "180")))
        try:
            if shared:
                torch.distributed.init_process_group("gloo", rank=rank, world_size=world, timeout=limit)
            else:
                torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=limit)
            backend = torch.distributed.get_backend()
            # first collective = the transport check: a failure (or a hang, cut by the timeout) surfaces HERE with a clear
            # message, not as a stalled timed region
            probe = torch.ones(1, device="cpu" if shared else device)
            torch.distributed.all_reduce(probe)
            if int(probe.item()) != world:
                raise RuntimeError("all_reduce over %d ranks returned %s" % (world, probe.item()))
            me = {"rank": rank, "host": socket.gethostname(), "device": dev_index, "pci": (placement or {}).get("pci"),
                  "numa_node": (placement or {}).get("numa_node")}
            identities = [None] * world
            torch.distributed.all_gather_object(identities, me)
        except Exception as e:
            raise SystemExit("rank %d/%d: process group (%s over %s:%s) could not be set up within %s: %s: %s — check that every rank sees "
                             "its GPU (HIP_VISIBLE_DEVICES), that HSA_ENABLE_IPC_MODE_LEGACY=0 is exported and that MASTER_ADDR resolves"
                             % (rank, world, "gloo" if shared else "nccl(RCCL)", os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"),
                                limit, type(e).__name__, str(e)[:300]))
        if not shared:
            seen = {(i["host"], i["device"]) for i in identities}
            if len(seen) != world:
                raise SystemExit("the %d ranks do not sit on %d distinct GPUs: %s" % (world, world, identities))

    from deftet_amd import _lib, sharding
    lib = _lib.load()
    gather = sharding.LossGather()
    wl = make_workload(args.config, rank, device, world, gather)
    elapsed, _, kms, kcnt = timed(wl, lib, args.steps, args.warmup, world)
    # median / maximum step: a second region of the same K steps with an event after every step (not the headline)
    _, per_step, _, _ = timed(wl, lib, args.steps, 2, world, step_events=True)
    rank_ms = None
    if world > 1:
        # the job's time is the slowest rank's; every rank's own time is kept beside it so that a straggler is visible
        t = torch.tensor([elapsed], device="cpu" if shared else device, dtype=torch.float64)
        allt = torch.empty(world, device=t.device, dtype=torch.float64)
        torch.distributed.all_gather_into_tensor(allt, t)
        rank_ms = [round(float(x) / args.steps * 1e3, 4) for x in allt.tolist()]
        elapsed = float(allt.max().item())

    if rank == 0:
        peak = measure_bandwidth(device) if (world == 1 and not args.no_bandwidth_probe) else None
        main_line = summarize(wl, elapsed, per_step, kms, kcnt, args.steps, world, peak)
        line = {
            "metric": ("M tet-point tests/s (fwd+bwd) at res=70, 100k queries" if args.config == 2 else
                       "%s (fwd+bwd), %s" % (wl.unit, CONFIGS[args.config]["name"])),
            "value": main_line["value"], "unit": main_line["unit"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_line["ms_per_step"], "ms_per_step_median": main_line["ms_per_step_median"],
            "ms_per_step_max": main_line["ms_per_step_max"],
            "median_max_from": "a second region of the same K steps with a HIP event after every step",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": wl.describe(),
            "roofline": main_line["roofline"],
        }
        if world > 1:
            line["rccl_ranks"] = torch.distributed.get_world_size()
            if line["rccl_ranks"] != args.gpus:
                raise SystemExit("process group has %d ranks, --gpus %d" % (line["rccl_ranks"], args.gpus))
            line["backend"] = backend
            line["ranks"] = identities                    # rank -> (host, device, PCI address, NUMA node its threads were pinned to)
            line["distinct_devices"] = len({(i["host"], i["device"]) for i in identities})
            line["ms_per_step_by_rank"] = {"min": min(rank_ms), "max": max(rank_ms), "all": rank_ms}
        if world == 1:
            if isinstance(wl, PitWorkload) and not args.no_unpipelined:
                # the same K steps with the other setting of the cross-step overlap (not the headline; printed beside it)
                wl.pipeline = not wl.pipeline
                e2, ps2, km2, kc2 = timed(wl, lib, args.steps, 2, 1, barrier=False, step_events=True)
                key = "pipelined" if wl.pipeline else "unpipelined"
                line["ms_per_step_%s" % key] = round(e2 / args.steps * 1e3, 4)
                line["ms_per_step_%s_median" % key] = round(statistics.median(ps2), 4)
                line["avg_launch_ms_%s" % key] = round(km2 / max(kc2, 1), 5)
                wl.pipeline = not wl.pipeline
                gms, why = graph_replay(wl, args.steps)
                line["ms_per_step_hipgraph"] = gms
                if why:
                    line["hipgraph_note"] = why
            if isinstance(wl, PitWorkload) and not args.no_brute_force:
                line["brute_force_hip"] = brute_force_comparator(wl, main_line["ms_per_step"])
            if not args.no_cpu_baseline and isinstance(wl, PitWorkload):
                line["cpu_baseline"] = cpu_baseline(wl)
            if not args.no_other_configs:
                del wl
                torch.cuda.empty_cache()
                others = []
                for cid in sorted(CONFIGS):
                    if cid == args.config:
                        continue
                    w2 = make_workload(cid, 0, device, 1, None)
                    k2 = max(20, args.steps)                      # (ten steps let the first, idle-GPU step weigh 17 % of the mean)
                    e, ps, km, kc = timed(w2, lib, k2, 3, 1, barrier=False, step_events=True)
                    rec = summarize(w2, e, ps, km, kc, k2, 1, peak)
                    entry = {"config_id": cid, "config": w2.describe()["workload"], "value": rec["value"], "unit": rec["unit"], "steps": k2,
                             "ms_per_step": rec["ms_per_step"], "ms_per_step_median": rec["ms_per_step_median"]}
                    if isinstance(w2, PitWorkload):
                        entry["ms_per_step_hipgraph"] = graph_replay(w2, k2)[0]
                    if isinstance(w2, RasterWorkload):
                        # both saturation policies every round (which one Kaolin implements is unknown; this configuration saturates)
                        entry["policy"] = ["nearest", "first"][w2.policy]
                        w2.policy = 1 - w2.policy
                        e3, ps3, _, _ = timed(w2, lib, k2, 3, 1, barrier=False, step_events=True)
                        entry["ms_per_step_%s" % ["nearest", "first"][w2.policy]] = round(e3 / k2 * 1e3, 4)
                        w2.policy = 1 - w2.policy
                        entry["saturated_pixels"] = w2.saturated()
                        entry["n_pixel"] = w2.P
                    if w2.dominant_bytes > 0:
                        entry["roofline"] = {k: rec["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "whole_step")}
                        if "valu" in rec["roofline"]:
                            entry["roofline"]["valu"] = rec["roofline"]["valu"]
                    elif "valu" in rec["roofline"]:
                        entry["roofline"] = dict(rec["roofline"]["valu"], kernel=rec["roofline"]["kernel"], avg_launch_ms=rec["roofline"]["avg_launch_ms"])
                    else:
                        entry["dominant_kernel"] = {"kernel": rec["roofline"]["kernel"], "avg_launch_ms": rec["roofline"]["avg_launch_ms"]}
                    others.append(entry)
                    del w2
                    torch.cuda.empty_cache()
                line["other_configs"] = others
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
