"""Which kernel of the point-in-tet forward pays for a degenerate query distribution (tools/probes/regime_probe.py): per-kernel times by the
library's own events.  python tools/probes/regime_kernels_probe.py"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deftet_amd import _lib, grids, hip_ops
lib = _lib.load(); dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
N = 100000
tet, pts, _, _ = grids.make_case(40, N, 2, 0.1)
T = torch.from_numpy(tet).to(dev)
u = torch.rand(2, N, 3, device=dev, generator=g) - 0.5
cases = {"uniform": u, "one tiny ball": u * 1e-3 + 0.1, "all identical": torch.zeros(2, N, 3, device=dev) + 0.25, "line": u * torch.tensor([1.0, 0.0, 0.0], device=dev),
         "ball of 1/20 of the cube": u * 0.05 + 0.1}
for name, q in cases.items():
    q = q.contiguous()
    out = {}
    for k in ("k_query_bbox", "k_slab_local", "k_slab_sort", hip_ops.pit_kernel_name(0, T.shape[1], N), "k_finalize"):
        hip_ops.point_in_tet(T, q, want_bary=True); torch.cuda.synchronize()
        lib.deftet_profile_select(k.encode())
        for _ in range(3): hip_ops.point_in_tet(T, q, want_bary=True)
        torch.cuda.synchronize()
        tot, cnt = ctypes.c_double(0), ctypes.c_longlong(0)
        lib.deftet_profile_read(ctypes.byref(tot), ctypes.byref(cnt)); lib.deftet_profile_select(b"")
        out[k] = round(tot.value / max(cnt.value, 1), 3)
    print("%-26s ms per kernel %s; grid %s" % (name, out, hip_ops.point_in_tet_grid(T.shape[1], N)), flush=True)
