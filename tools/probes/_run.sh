cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3s
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3s/kt -o kt --output-format csv -- python $GRAFT_REPO_ROOT/tools/step_demo.py --surface --steps 5 > $GRAFT_REPO_ROOT/gpurun_out/r3s/kt.log 2>&1)
tail -1 gpurun_out/r3s/kt.log | cut -c1-300
python - $(find gpurun_out/r3s/kt -name "*kernel_stats.csv" | head -1) <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print('total kernel time per step (7 steps incl warmup): %.2f ms, launches per step: %d' % (tot/7/1e6, calls/7))
for r in rows[:28]:
    print('  ',r['Name'].split('(')[0][-50:].ljust(52), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
rm -rf gpurun_out/r3s/kt
