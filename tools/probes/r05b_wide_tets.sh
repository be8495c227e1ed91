cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 200 python tools/probes/regime_probe.py 2>&1 | tail -9
timeout 400 python -m pytest tests/test_point_in_tet_gpu.py -x -q 2>&1 | tail -2
FUZZ_SEED=901 timeout 100 python tools/probes/fuzz_point_in_tet.py 60 2>&1 | tail -1
for rep in 1 2; do for n in prev product; do lib=""; [ "$n" != product ] && lib="DEFTET_HIP_LIB=$PWD/tools/probes/bin/libdeftet_$n.so"; for c in 2 3 1; do echo -n "$n "; env $lib python tools/probes/sort_probe.py --config $c 2>/dev/null | tail -1; done; done; done
