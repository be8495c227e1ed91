#!/usr/bin/env python
"""Where the order rule's crossover lies: the Kuhn res-70 tet list with a fraction of its positions shuffled among themselves —
far-step fraction (tet_order_coherence) of the caller's numbering and of the computed order, and the traversal's time inside the
rotating-set step with either.  hip_ops._FAR_LIMIT is set where "sorted" starts to win.  One JSON line per fraction."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "probes"))
import bench  # noqa: E402
import scan_variants  # noqa: E402
from deftet_amd import _lib, hip_ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    fracs = [float(x) for x in (sys.argv[1:] or ["0", "0.01", "0.02", "0.04", "0.08", "0.15", "0.3", "1.0"])]
    for f in fracs:
        rec = {"shuffle_frac": f}
        for mode in ("native", "sorted"):
            cfg = dict(bench.CONFIGS[2], sets=3, mesh="shuffled", shuffle_frac=f, tet_order=mode)
            wl = bench.PitWorkload(cfg, 0, dev, 1, None, pipeline=False)
            if mode == "native":
                t0 = wl.sets[0]["tet"][0]
                order = hip_ops.tet_spatial_order(t0)
                a, b = hip_ops.tet_order_coherence(t0).tolist(), hip_ops.tet_order_coherence(t0, order).tolist()
                rec["far_native"], rec["far_sorted"] = round(a[0] / a[1], 4), round(b[0] / b[1], 4)
                rec["rule_says"] = "sorted" if hip_ops._decide_order(a[0] / a[1], b[0] / b[1]) else "native"
            k_us, step_us = scan_variants.run_steplike(wl, lib, 0, 20, hip_ops.pit_kernel_name(0, wl.T, wl.Q))
            rec["traversal_us_" + mode], rec["step_us_" + mode] = round(k_us, 1), round(step_us, 1)
            del wl
            torch.cuda.empty_cache()
        rec["faster"] = "sorted" if rec["traversal_us_sorted"] < rec["traversal_us_native"] else "native"
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
