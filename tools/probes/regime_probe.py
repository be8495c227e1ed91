"""Timing of the grid-accelerated operators OUTSIDE the training distribution they were tuned on: clustered, planar, collinear, far and
duplicated inputs at 100 k points.  A regime that costs 100x the uniform case is a cliff worth knowing.  python tools/probes/regime_probe.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deftet_amd import grids, hip_ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)

def timeit(f, n=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

N = 100000
def clouds(B=2):
    u = torch.rand(B, N, 3, device=dev, generator=g) - 0.5
    sph = torch.nn.functional.normalize(torch.randn(B, N, 3, device=dev, generator=g), dim=-1) * 0.4
    out = {"uniform": u, "sphere surface": sph, "one tiny ball": u * 1e-3 + 0.1, "plane z=0": u * torch.tensor([1.0, 1.0, 0.0], device=dev),
           "line": u * torch.tensor([1.0, 0.0, 0.0], device=dev), "two far clusters": torch.where((torch.arange(N, device=dev) % 2 == 0)[None, :, None], u * 0.01 - 5.0, u * 0.01 + 5.0),
           "all identical": torch.zeros(B, N, 3, device=dev) + 0.25, "half duplicates": torch.cat([u[:, : N // 2], u[:, : N // 2]], 1)}
    return {k: v.contiguous() for k, v in out.items()}

C = clouds()
print("A10 nn_index (100k queries x 100k points, 2 shapes): ms", flush=True)
for qn in ("uniform", "sphere surface"):
    for pn, pc in C.items():
        t = timeit(lambda: hip_ops.nn_index(C[qn], pc))
        print("  queries %-15s points %-16s %9.3f" % (qn, pn, t), flush=True)
# A9: points against a triangulated sphere (res-40 marching surface is not at hand: an icosphere-like fan of ~80k faces)
th = torch.rand(80000, device=dev, generator=g) * 6.283; ph = torch.acos(2 * torch.rand(80000, device=dev, generator=g) - 1)
c = 0.4 * torch.stack([torch.sin(ph) * torch.cos(th), torch.sin(ph) * torch.sin(th), torch.cos(ph)], -1)
tri = (c[:, None, :] + 0.01 * torch.randn(80000, 3, 3, device=dev, generator=g))[None].expand(2, -1, -1, -1).contiguous()
nf = torch.full((2,), 80000.0, device=dev)
print("A9 tri_dist_fwd (100k points x 80k small triangles on a sphere, 2 shapes): ms", flush=True)
for pn, pc in C.items():
    t = timeit(lambda: hip_ops.tri_dist_fwd(pc, tri, nf))
    print("  points %-16s %9.3f" % (pn, t), flush=True)
big = tri.clone(); big[:, :200] *= 40.0                      # 200 huge triangles among the small ones
t = timeit(lambda: hip_ops.tri_dist_fwd(C["sphere surface"], big, nf)); print("  points sphere surface, 200 of the faces 40x larger %9.3f" % t, flush=True)
# A1 forward with clustered queries
tet, pts, _, _ = grids.make_case(40, N, 2, 0.1)
T = torch.from_numpy(tet).to(dev)
print("A1 point_in_tet forward (res-40 grid, 48,000 tets, 100k queries, 2 shapes): ms", flush=True)
for pn, pc in C.items():
    t = timeit(lambda: hip_ops.point_in_tet(T, pc, want_bary=True))
    print("  queries %-16s %9.3f" % (pn, t), flush=True)
