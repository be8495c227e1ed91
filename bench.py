#!/usr/bin/env python
"""bench.py — headline benchmark of the DefTet per-tetrahedron hot path on MI355X.

Metric (BASELINE.json): M tet-point tests/s (fwd+bwd) at res=70, 100k queries.
One "step" = one pass of the hot path over one batch of B=8 synthetic shapes per GPU:
    fwd : point-in-tet index (A1) + barycentric weights of the hit tet + paste_occ gather
    bwd : dL/dtet scatter (A1b) + dL/dpred scatter (paste_occ backward)
`value` counts NOMINAL tet-point pairs B*T*Q per step (what the reference's brute-force
kernel enumerates), inputs resident in HBM, acceleration-structure build included.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RES, N_QUERY, BATCH = 70, 100_000, 8          # BASELINE.json configs[2]
HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: 8 TB/s spec


def make_inputs(rank, device, res=RES, n_query=N_QUERY, batch=BATCH):
    from deftet_amd import grids
    verts, tets = grids.kuhn_grid(res)
    pos = grids.jittered_positions(verts, res, batch, 0.1, seed0=1000 + rank * batch)
    tet = grids.gather_tets(pos, tets)
    pts = grids.random_queries(batch, n_query, seed0=2000 + rank * batch)
    gw = np.stack([np.random.default_rng(4000 + rank * batch + b).standard_normal((n_query, 4)).astype(np.float32)
                   for b in range(batch)])
    pred = np.stack([np.random.default_rng(5000 + rank * batch + b).random(tet.shape[1]).astype(np.float32)
                     for b in range(batch)])
    gout = np.stack([np.random.default_rng(6000 + rank * batch + b).standard_normal(n_query).astype(np.float32)
                     for b in range(batch)])
    host = dict(tet=tet, pts=pts)
    dev = {k: torch.from_numpy(v).to(device) for k, v in dict(tet=tet, pts=pts, gw=gw, pred=pred, gout=gout).items()}
    return host, dev


ALGO = int(os.environ.get("DEFTET_BENCH_ALGO", "0"))      # A/B switch for the traversal kernel (0 = shipped default)
DOMINANT = {0: b"k_tet_scan", 2: b"k_tet_scan_staged", 3: b"k_tet_scan_rows"}.get(ALGO, b"k_tet_scan")


GATHER = None                    # sharding.LossGather(), created in main() once the process group exists
PIPELINE = os.environ.get("DEFTET_BENCH_PIPELINE", "1") not in ("", "0")
OVERLAP_WITH = os.environ.get("DEFTET_BENCH_OVERLAP", "fwd")      # "fwd": start at once (measured 0.262 ms/step); "bwd": after this step's forward (0.276)
_SIDE = None


def side_stream():
    global _SIDE
    if _SIDE is None:
        _SIDE = torch.cuda.Stream()
    return _SIDE


def step(d, world):
    """fwd: index + weights + fused paste_occ gather (+ per-tet hit records); bwd: dL/dtet and
    dL/dpred from one per-tet pass over those records (no atomics); then the per-shape loss
    scalars (all-gathered when world > 1).  Same calls as the PointInTetOcc autograd op makes."""
    from deftet_amd import hip_ops, sharding
    if PIPELINE:
        # software pipelining across steps: the query side of the operator (bounding box + counting sort,
        # five small latency-bound kernels that need only the points) is enqueued on a second stream as
        # soon as the previous step's traversal has been launched, so it overlaps with that step's big
        # kernels; every step still does all of its work (K sorts in the K timed steps)
        pq = d.pop("pq", None)
        if pq is None:
            pq = hip_ops.prepare_queries(d["pts"], d["tet"].shape[1], algo=ALGO)
        cond, w, occ, hits = hip_ops.point_in_tet(d["tet"], d["pts"], want_bary=True, pred_bxt=d["pred"], want_hits=True, algo=ALGO,
                                                  prepared=pq)
        if OVERLAP_WITH == "bwd":                                     # start it when this step's forward has finished
            side_stream().wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side_stream()):
            d["pq"] = hip_ops.prepare_queries(d["pts"], d["tet"].shape[1], algo=ALGO)
    else:
        cond, w, occ, hits = hip_ops.point_in_tet(d["tet"], d["pts"], want_bary=True, pred_bxt=d["pred"], want_hits=True, algo=ALGO)
    g_tet, _, g_pred = hip_ops.point_in_tet_bwd(d["tet"], d["pts"], cond, d["gw"], grad_occ=d["gout"], hits=hits)
    loss = hip_ops.rowdot(w, d["gw"], occ, d["gout"])         # [B] per-shape loss scalars
    if world > 1:
        # the only collective (RCCL over xGMI): enqueued asynchronously, consumed one step later, so
        # the next step's kernels do not wait for it; main() flushes the last one inside the timed region
        if torch.distributed.get_backend() == "gloo":                         # single-GPU test hook: stage through the host
            loss = GATHER.submit(loss.cpu())
        else:
            loss = GATHER.submit(loss)
    return cond, w, g_tet, g_pred, loss


def cpu_baseline(host):
    """Oracle (CPU restatement, kind="port") on a bounded sample of the same workload."""
    from oracle import oracle as O
    tet = host["tet"][:1]
    ncpu = os.cpu_count() or 1
    q1 = 1500
    t0 = time.perf_counter()
    O.point_in_tet(tet, host["pts"][:1, :q1])
    t1 = time.perf_counter() - t0
    qn = min(N_QUERY, max(2000, 1200 * ncpu))
    t0 = time.perf_counter()
    _, nthreads = O.point_in_tet(tet, host["pts"][:1, :qn], omp=True, return_executed=True)
    tn = time.perf_counter() - t0
    T = tet.shape[1]
    return {
        "value": round(T * qn / tn / 1e6, 2), "unit": "M tet-point tests/s (fwd only)", "cores": int(nthreads),
        "kind": "port",
        "sample": "oracle/deftet_oracle.c brute-force scan, 1 shape res=%d (T=%d), first %d queries, OpenMP over "
                  "queries; single-core on %d queries: %.2f M/s" % (RES, T, qn, q1, T * q1 / t1 / 1e6),
        "value_1core": round(T * q1 / t1 / 1e6, 2),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if shared:
            torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from deftet_amd import _lib
    lib = _lib.load()
    host, d = make_inputs(rank, device)
    B, T, Q = d["tet"].shape[0], d["tet"].shape[1], d["pts"].shape[1]

    global GATHER
    from deftet_amd import sharding
    GATHER = sharding.LossGather()
    for _ in range(args.warmup):
        step(d, world)
    if world > 1:
        GATHER.flush()
    torch.cuda.synchronize()

    dominant = DOMINANT
    lib.deftet_profile_select(dominant)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(d, world)
    if world > 1:
        GATHER.flush()                                   # the last step's all-gather belongs to the timed region
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    tot_ms, cnt = ctypes.c_double(0), ctypes.c_longlong(0)
    lib.deftet_profile_read(ctypes.byref(tot_ms), ctypes.byref(cnt))
    lib.deftet_profile_select(b"")

    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if shared else device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        pairs = float(world) * B * T * Q * args.steps
        kern_ms = tot_ms.value / max(cnt.value, 1)
        # SURVEY.md 8(d), A1 fwd: B*(48*T + 12*Q + 4*Q) algorithmic bytes per call — exactly what k_tet_scan must
        # touch once (48-byte tet records, 16-byte sorted queries).  Its 16-byte-per-tet hit records and the
        # result atomics are overhead of THIS design and are not counted.
        algo_bytes = B * (48.0 * T + 16.0 * Q)
        achieved = algo_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        # whole step: fwd with weights B*(48T+16Q+16Q), bwd B*(32Q + 48*hits + 48T), hits ~ 0.864*Q
        step_bytes = B * (48.0 * T + 32.0 * Q) + B * (32.0 * Q + 48.0 * 0.864 * Q + 48.0 * T)
        step_gbs = step_bytes / (elapsed / args.steps) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("k_tet_scan_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "M tet-point tests/s (fwd+bwd) at res=70, 100k queries",
            "value": round(pairs / elapsed / 1e6, 1),
            "unit": "M tet-point tests/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: res=70 Kuhn tet grid (T=%d), %d uniform queries, batch=%d shapes "
                                   "per GPU, point-in-tet index + weights + paste_occ, fwd+bwd, grid build included" % (T, Q, B),
                       "res": RES, "n_tet": T, "n_query": Q, "batch_per_gpu": B,
                       "sharding": "shapes sharded by rank; all-gather of %d loss scalars" % (world * B),
                       "pipelining": ("query sort of step i+1 enqueued on a second stream during step i" if PIPELINE else "none")},
            "roofline": {"bound": "hbm", "kernel": dominant.decode(), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": round(kern_ms, 5),
                         "launches_timed": int(cnt.value),
                         "whole_step": {"algorithmic_bytes": step_bytes, "achieved": round(step_gbs, 1),
                                        "frac": round(step_gbs / HBM_PEAK_GBS, 4)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(host)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
