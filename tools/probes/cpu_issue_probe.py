"""CPU-side cost of issuing one bench step (no synchronisation inside the loop) next to its GPU time, and the same per
hip_ops call: is the eager loop of bench.py launch-bound?   python tools/probes/cpu_issue_probe.py [--config 2]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from deftet_amd import hip_ops

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--steps", type=int, default=200)
a = ap.parse_args()
wl = bench.make_workload(a.config, 0, torch.device("cuda:0"), 1)
if wl is None:
    raise SystemExit("bench.make_workload missing")
for i in range(5):
    wl.step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    wl.step(i)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
d = wl.sets[0]
out = {"config": a.config, "steps": a.steps, "cpu_issue_us_per_step": round(t_issue / a.steps * 1e6, 1),
       "wall_us_per_step": round(t_all / a.steps * 1e6, 1)}
# per call (GPU drained first so that nothing blocks on a full queue)
def cpu_us(fn, n=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return round(dt / n * 1e6, 1), r
out["point_in_tet_us"], r = cpu_us(lambda: hip_ops.point_in_tet(d["tet"], d["pts"], want_bary=True, pred_bxt=d["pred"], want_hits=True, algo=wl.algo), 50)
cond, w, occ, hits = r
out["point_in_tet_bwd_us"], _ = cpu_us(lambda: hip_ops.point_in_tet_bwd(d["tet"], d["pts"], cond, d["gw"], grad_occ=d["gout"], hits=hits), 50)
out["rowdot_us"], _ = cpu_us(lambda: hip_ops.rowdot(w, d["gw"], occ, d["gout"]), 50)
print(json.dumps(out))
