"""Turns two rocprofv3 counter passes into profiles/pmc_traffic.json.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch -o f --output-format csv -- \
        python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-bandwidth-probe
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write -o w --output-format csv -- \
        python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-bandwidth-probe
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/pmc_traffic.json

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in
KiB; on gfx950 FETCH_SIZE reports half of a coalesced stream (calibrated on this box: k_query_bbox
reads 9.6 MB of points and reports 4.8 MB), WRITE_SIZE is exact.  Values are averages per launch.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def per_kernel(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no *counter_collection.csv under %s" % d)
    tot, cnt = defaultdict(float), defaultdict(int)
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
        tot[name] += float(row["Counter_Value"]) * 1024.0
        cnt[name] += 1
    return {k: tot[k] / cnt[k] for k in tot}, dict(cnt)


def source_sha1():
    """sha1 of the kernel source the counters belong to: bench.py prints the traffic figure only while this still matches."""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return hashlib.sha1(open(os.path.join(root, "deftet_amd", "csrc", "point_in_tet.hip"), "rb").read()).hexdigest()


def main(fetch_dir, write_dir):
    (f, fc), (w, _) = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    out = {"note": __doc__.split("Corrections", 1)[1].strip().replace("\n", " "), "kernel_source_sha1": source_sha1(),
           "commit": os.environ.get("DEFTET_COMMIT", "(fill in: git rev-parse HEAD of the measured tree)"), "per_kernel": {}}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("deftet::"):
            continue
        out["per_kernel"][k] = {"FETCH_SIZE_bytes_raw": round(f.get(k, 0.0)), "fetch_bytes_corrected_x2": round(2 * f.get(k, 0.0)),
                                "WRITE_SIZE_bytes": round(w.get(k, 0.0))}
    for k, v in out["per_kernel"].items():                      # "<short kernel name>_hbm_bytes_per_launch" (bench.py reads the dominant kernel's)
        short = re.sub(r"<.*", "", k.split("::")[-1])              # template instances share their kernel's key (k_tet_scan_wave<false>)
        out["%s_hbm_bytes_per_launch" % short] = out.get("%s_hbm_bytes_per_launch" % short, 0) or (v["fetch_bytes_corrected_x2"] + v["WRITE_SIZE_bytes"])
    # the step = the kernels launched once per step: everything launched at least half as often as the most frequent kernel
    # (the one-time launches of a run — the traversal-order choice of the first call, the first call's measured query box —
    # are in per_kernel but not in the sum)
    most = max((fc.get(k, 0) for k in out["per_kernel"]), default=0)
    step_kernels = [k for k in out["per_kernel"] if fc.get(k, 0) * 2 >= most]
    for k, v in out["per_kernel"].items():
        v["launches"] = fc.get(k, 0)
    out["step_kernels"] = [k.split("::")[-1] for k in step_kernels]
    out["whole_step_hbm_bytes"] = sum(out["per_kernel"][k]["fetch_bytes_corrected_x2"] + out["per_kernel"][k]["WRITE_SIZE_bytes"] for k in step_kernels)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:3])
