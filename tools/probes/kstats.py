"""Per-kernel summary (calls, avg us) from a rocprofv3 --kernel-trace csv directory."""
import csv, glob, os, re, sys
from collections import defaultdict
files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
tot, cnt = defaultdict(float), defaultdict(int)
for row in csv.DictReader(open(files[0])):
    name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
    name = re.sub(r"<.*", "<>", name)
    tot[name] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    cnt[name] += 1
for k in sorted(tot, key=lambda k: -tot[k])[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print("%-90s %6d %10.1f us avg %12.1f us total" % (k[:90], cnt[k], tot[k] / cnt[k], tot[k]))
