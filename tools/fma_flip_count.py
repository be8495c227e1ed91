#!/usr/bin/env python
"""How many point-in-tet decisions does fused multiply-add contraction flip?  (reference
check_condition_tet_for.cu:105-121 was compiled by nvcc with its default --fmad=true, the parity oracle follows the source
text operation by operation.)  Runs the brute-force CPU scan twice — oracle/libdeftet_oracle.so (-ffp-contract=off) and
oracle/libdeftet_oracle_fma.so (-mfma -ffp-contract=fast, same source) — on BASELINE configs[2]'s data and prints one
JSON line with the number of queries whose answer differs.  CPU only; a few minutes on 128 cores.

    python tools/fma_flip_count.py [--shapes 2] [--queries 100000] [--res 70]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deftet_amd import grids  # noqa: E402
from oracle import oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", type=int, default=2)
ap.add_argument("--queries", type=int, default=100_000)
ap.add_argument("--res", type=int, default=70)
a = ap.parse_args()
tet, pts, _, _ = grids.make_case(a.res, a.queries, a.shapes)
t0 = time.time()
plain = O.point_in_tet(tet, pts, omp=True)
t1 = time.time()
fused = O.point_in_tet_contracted(tet, pts)
t2 = time.time()
diff = plain != fused
hit_to_miss = int(((plain >= 0) & (fused < 0)).sum())
miss_to_hit = int(((plain < 0) & (fused >= 0)).sum())
print(json.dumps({"res": a.res, "n_tet": int(tet.shape[1]), "shapes": a.shapes, "queries_per_shape": a.queries,
                  "decisions": int(plain.size), "flipped": int(diff.sum()), "flip_rate": float(diff.mean()),
                  "hit_to_miss": hit_to_miss, "miss_to_hit": miss_to_hit, "other_tet": int(diff.sum()) - hit_to_miss - miss_to_hit,
                  "cores": os.cpu_count(), "seconds_plain": round(t1 - t0, 1), "seconds_fused": round(t2 - t1, 1)}))
