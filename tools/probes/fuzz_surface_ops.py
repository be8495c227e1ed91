"""Fuzz: accelerated exact paths against the streaming-scan / brute kernels of the same library on random
inputs — check_sign (binned vs brute), nn_index (grid vs scan), tri_dist_fwd (grid vs scan), face_edge_adj
(sort vs O(F^2)).   python tools/probes/fuzz_surface_ops.py [seconds]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deftet_amd import hip_ops

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "2")))
g = torch.Generator(device=dev).manual_seed(11)
t_end = time.time() + budget
n = 0


def rnd(*shape):
    return torch.rand(*shape, device=dev, generator=g) - 0.5


def spoil(x, frac=0.02):
    m = torch.rand(x.shape[:-1], device=dev, generator=g)
    x = x.clone()
    x[m < frac] = float("nan")
    x[(m > frac) & (m < 2 * frac)] = float("inf")
    x[(m > 2 * frac) & (m < 3 * frac)] *= 4e6
    return x


while time.time() < t_end:
    B = int(rng.integers(1, 4))
    scale = float(rng.choice([1e-2, 1.0, 1.0, 30.0]))
    dirty = rng.random() < 0.3
    # ---- check_sign
    V, F, N = int(rng.choice([3, 10, 200, 2000])), int(rng.choice([1, 2, 50, 700, 5000])), int(rng.choice([1, 64, 65, 1000, 20000]))
    verts = rnd(B, V, 3) * scale
    if rng.random() < 0.3:
        verts[..., 1] = verts[..., 1].round(decimals=1)            # many faces edge-on to the ray
    faces = torch.randint(0, V, (F, 3), device=dev, generator=g)
    pts = rnd(B, N, 3) * scale * 1.2
    if dirty:
        verts, pts = spoil(verts), spoil(pts)
    a, ca = hip_ops.check_sign(verts, faces, pts, return_count=True, check=False)
    b, cb = hip_ops.check_sign(verts, faces, pts, brute=True, return_count=True, check=False)
    if not torch.equal(ca, cb):
        print("check_sign MISMATCH B=%d V=%d F=%d N=%d scale=%g dirty=%s" % (B, V, F, N, scale, dirty), flush=True); sys.exit(1)
    # ---- nn_index
    Nq, M = int(rng.choice([1, 63, 1000, 9000])), int(rng.choice([1, 2, 100, 5000, 40000]))
    q, p = rnd(B, Nq, 3) * scale, rnd(B, M, 3) * scale * float(rng.choice([0.05, 1.0]))
    if rng.random() < 0.3:
        p[..., 2] = 0.0
    if rng.random() < 0.3:
        p[:, : M // 2] = p[:, M // 2: M // 2 * 2]                 # duplicates: lowest index must win
    if dirty:
        q, p = spoil(q), spoil(p, 0.005)
    if not torch.equal(hip_ops.nn_index(q, p), hip_ops.nn_index(q, p, brute=True)):
        print("nn_index MISMATCH B=%d N=%d M=%d scale=%g dirty=%s" % (B, Nq, M, scale, dirty), flush=True); sys.exit(1)
    # ---- tri_dist_fwd
    P, Ft = int(rng.choice([1, 64, 700, 6000])), int(rng.choice([1, 3, 100, 2500]))
    tp = rnd(B, P, 3) * scale
    cen = rnd(B, Ft, 1, 3) * scale
    tf = cen + rnd(B, Ft, 3, 3) * scale * float(rng.choice([0.01, 0.1, 1.0]))
    if dirty:
        tp, tf = spoil(tp), spoil(tf.reshape(B, Ft * 3, 3), 0.004).reshape(B, Ft, 3, 3)
    nfb = torch.tensor([float(max(1, Ft - int(rng.integers(0, min(3, Ft))))) for _ in range(B)], device=dev)
    d1, f1 = hip_ops.tri_dist_fwd(tp, tf, nfb)
    d2, f2 = hip_ops.tri_dist_fwd(tp, tf, nfb, brute=True)
    same = torch.equal(f1, f2) and torch.equal(torch.nan_to_num(d1, nan=-7.0), torch.nan_to_num(d2, nan=-7.0))
    if not same:
        print("tri_dist MISMATCH B=%d P=%d F=%d scale=%g dirty=%s" % (B, P, Ft, scale, dirty), flush=True); sys.exit(1)
    # ---- face_edge_adj
    Fe = int(rng.choice([1, 2, 40, 900]))
    vv = (rnd(max(3, Fe // 2), 3) * scale)
    fe = vv[torch.randint(0, vv.shape[0], (Fe, 3), device=dev, generator=g)]
    if dirty:
        fe = spoil(fe.reshape(-1, 3), 0.01).reshape(Fe, 3, 3)
    if not torch.equal(hip_ops.face_edge_adj(fe), hip_ops.face_edge_adj(fe, brute=True)):
        print("face_edge_adj MISMATCH F=%d scale=%g dirty=%s" % (Fe, scale, dirty), flush=True); sys.exit(1)
    n += 1
print("fuzz ok: %d random rounds of four operators" % n, flush=True)
