#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc passes (any number of output directories) -> one JSON document.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU ... --kernel-trace -d $R/gpurun_out/pmcA -o a --output-format csv -- python ...
    python tools/pmc_summary.py gpurun_out/pmcA gpurun_out/pmcB ... [--match k_tet_scan] > profiles/xyz.json
Counter values are averaged per launch of each kernel; FETCH_SIZE / WRITE_SIZE are converted to bytes (KiB units) and
FETCH_SIZE is also shown x2 (the gfx950 correction of MI355X_MICROARCH.md, see tools/pmc_traffic.py)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def main(argv):
    match = None
    dirs = []
    it = iter(argv)
    for a in it:
        if a == "--match":
            match = next(it)
        else:
            dirs.append(a)
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
                if match and match not in name:
                    continue
                c = acc[name][row["Counter_Name"]]
                c[0] += float(row["Counter_Value"])
                c[1] += 1
    out = {}
    for k in sorted(acc):
        rec = {}
        for c, (tot, n) in sorted(acc[k].items()):
            v = tot / max(n, 1)
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                rec[c + "_bytes"] = round(v * 1024)
                if c == "FETCH_SIZE":
                    rec["FETCH_SIZE_bytes_x2"] = round(v * 2048)
            else:
                rec[c] = round(v)
            rec.setdefault("_launches", n)
        out[k] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
