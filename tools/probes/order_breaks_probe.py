#!/usr/bin/env python
"""Coherence counts of deftet_tet_spatial_order_f32 (breaks of the caller's order / of the computed order inside groups of 64
consecutive tets) for the tet numberings the round-6 rule of hip_ops.auto_tet_order was fitted on.  One JSON line per case."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deftet_amd import grids, hip_ops  # noqa: E402


def cases():
    for res in (20, 40, 70):
        verts, tets = grids.kuhn_grid(res)
        pos = grids.jittered_positions(verts, res, 1, 0.1, seed0=1)
        yield "kuhn%d" % res, grids.gather_tets(pos, tets)[0]
        n = res // 2
        idx = np.arange(n * n * n * 6).reshape(n, n, n, 6)
        yield "kuhn%d/xfast" % res, grids.gather_tets(pos, tets[idx.transpose(2, 1, 0, 3).reshape(-1)])[0]
        yield "kuhn%d/yfast" % res, grids.gather_tets(pos, tets[idx.transpose(0, 2, 1, 3).reshape(-1)])[0]
        yield "kuhn%d/shuffled" % res, grids.gather_tets(pos, tets[np.random.default_rng(7).permutation(tets.shape[0])])[0]
        # blocks of 512 consecutive tets kept, the blocks shuffled: locally coherent, globally not
        nb = tets.shape[0] // 512
        perm = (np.random.default_rng(8).permutation(nb)[:, None] * 512 + np.arange(512)[None]).reshape(-1)
        yield "kuhn%d/blocks512" % res, grids.gather_tets(pos, tets[perm])[0]
    g40 = np.load(os.path.join(ROOT, "tests", "golden", "cube40_grid.npz"))
    v, t = g40["verts"].astype(np.float32), g40["tets"]
    yield "cube40", np.ascontiguousarray(v[t])
    yield "cube40/shuffled", np.ascontiguousarray(v[t[np.random.default_rng(7).permutation(t.shape[0])]])


def main():
    dev = torch.device("cuda:0")
    for name, tet in cases():
        t = torch.from_numpy(np.ascontiguousarray(tet, dtype=np.float32)).to(dev)
        _, breaks = hip_ops.tet_spatial_order(t, want_breaks=True)
        native, srt = breaks.tolist()
        T = t.shape[0]
        pairs = T - (T + 63) // 64
        print(json.dumps({"case": name, "n_tet": T, "breaks_native": native, "breaks_sorted": srt,
                          "frac_native": round(native / max(pairs, 1), 4), "frac_sorted": round(srt / max(pairs, 1), 4)}), flush=True)


if __name__ == "__main__":
    main()
