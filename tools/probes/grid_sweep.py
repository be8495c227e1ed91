import itertools, json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools", "probes"))
import torch, bench, scan_variants as sv
from deftet_amd import _lib, hip_ops
lib = _lib.load(); dev = torch.device("cuda:0")
for cfgid in [int(a) for a in sys.argv[1:]] or (2, 3, 1):
    cfg = dict(bench.CONFIGS[cfgid], sets=1)
    wl = bench.PitWorkload(cfg, 0, dev, 1, None, pipeline=False)
    sv.set_env(DEFTET_PIT_XFINE=None, DEFTET_PIT_GDIV=None, DEFTET_PIT_QDIV=None)
    ref, gref, *_ = sv.run(wl, lib, 0, 2)
    for (xf, gd, qd), algo in itertools.product([(6, 6, 2), (6, 4, 1.3), (6, 3, 1), (6, 2, 0.7), (6, 1.5, 0.5), (4, 3, 1), (8, 3, 1), (4, 1.5, 0.5), (3, 1.5, 0.5), (6, 9, 3)], (4,)):
        sv.set_env(DEFTET_PIT_XFINE=xf, DEFTET_PIT_GDIV=gd, DEFTET_PIT_QDIV=qd)
        outs, g, k_us, fwd_us, bwd_us = sv.run(wl, lib, algo, 6)
        print(json.dumps(dict(config=cfgid, algo=algo, xfine=xf, gdiv=gd, qdiv=qd, traversal_us=round(k_us, 1), fwd_us=round(fwd_us, 1), same=bool(torch.equal(outs[0], ref[0])))), flush=True)
    del wl
