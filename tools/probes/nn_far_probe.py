"""nn_index, far-query case of tools/bench_ops.py (uniform queries in the cube, 100k points on a sphere)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from deftet_amd import hip_ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
d = rng.normal(size=(100000, 3))
gt = (0.3 * d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
gt_d = torch.from_numpy(gt).to(dev)[None]
if len(sys.argv) > 1 and sys.argv[1] == "train":        # 20 samples per boundary face of a sphere-like surface (deftet.py:174-177)
    from tests.test_surface_ops_gpu import sphere_surface
    from deftet_amd import surface_losses
    face_d = torch.from_numpy(sphere_surface(70)).to(dev)
    q_d = surface_losses.sample_on_faces(face_d[None], 20).reshape(1, -1, 3).contiguous()
else:
    q_d = torch.from_numpy(rng.uniform(-0.3, 0.3, (1, 80640, 3)).astype(np.float32)).to(dev)
for name, fn in (("grid", lambda: hip_ops.nn_index(q_d, gt_d)), ("brute", lambda: hip_ops.nn_index(q_d, gt_d, brute=True))):
    r = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        r = fn()
    e1.record()
    torch.cuda.synchronize()
    print(name, "ms", round(e0.elapsed_time(e1) / 3, 3))
if not os.environ.get('NN_NO_CHECK'):
    assert torch.equal(hip_ops.nn_index(q_d, gt_d), hip_ops.nn_index(q_d, gt_d, brute=True))

from deftet_amd import _lib
lib = _lib.load()
if hasattr(lib, "deftet_debug_nn_stats"):
    import ctypes as C
    out = (C.c_ulonglong * 16)()
    lib.deftet_debug_nn_stats(out, 1)
    hip_ops.nn_index(q_d, gt_d)
    torch.cuda.synchronize()
    lib.deftet_debug_nn_stats(out, 1)
    v = list(out)
    waves = v[3]
    print("waves", waves, "records per wave: A %.0f B %.0f C %.0f; stream calls per wave: A %.1f B %.1f C %.1f" % (
        v[0] / waves, v[1] / waves, v[2] / waves, v[4] / waves, v[5] / waves, v[6] / waves))
    print("s_memtime ticks (100 MHz) per wave, mean / max: A %.0f / %d, B %.0f / %d, C %.0f / %d, final wait %.0f / %d" % (
        v[8] / waves, v[12], v[9] / waves, v[13], v[10] / waves, v[14], v[11] / waves, v[15]))
