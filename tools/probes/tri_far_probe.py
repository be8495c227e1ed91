"""tri_dist_fwd when the points are not close to the surface (early training): accelerated path vs streaming scan."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from deftet_amd import hip_ops
from tests.test_surface_ops_gpu import sphere_surface
dev = torch.device("cuda:0")
face = sphere_surface(70)
F = face.shape[0]
rng = np.random.default_rng(3000)
d = rng.standard_normal((100000, 3))
unit = d / np.linalg.norm(d, axis=1, keepdims=True)
face_d = torch.from_numpy(face).to(dev)
nfb = torch.tensor([float(F)], device=dev)
r_face = float(np.linalg.norm(face.reshape(-1, 3), axis=1).mean())
print("faces", F, "mean radius of the surface %.3f" % r_face)
for scale in ([float(a) for a in sys.argv[1:]] or (1.0, 1.1, 1.3, 2.0, 0.5, 0.05)):
    gt = torch.from_numpy((r_face * scale * unit).astype(np.float32)).to(dev)[None]
    res = {}
    for name, kw in (("grid", {}), ("scan", {"brute": True})):
        fn = lambda: hip_ops.tri_dist_fwd(gt, face_d[None], nfb, **kw)
        out = fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            out = fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / 3, out)
    same = torch.equal(res["grid"][1][0], res["scan"][1][0]) and torch.equal(res["grid"][1][1], res["scan"][1][1])
    print("points at %.2f x the surface radius: grid %.3f ms, scan %.3f ms, identical %s" % (scale, res["grid"][0], res["scan"][0], same))

