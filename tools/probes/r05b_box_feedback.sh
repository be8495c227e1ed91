set -x
mkdir -p gpurun_out/r05b
cd /root/repo
timeout 300 python -m pytest tests/test_point_in_tet_gpu.py -x -q -k "misses or backs_off or query_box" > gpurun_out/r05b/box_tests.log 2>&1; echo "rc $?" >> gpurun_out/r05b/box_tests.log
tail -5 gpurun_out/r05b/box_tests.log
timeout 900 python -m pytest tests/test_point_in_tet_gpu.py tests/test_pipeline_gpu.py -x -q > gpurun_out/r05b/pit_tests.log 2>&1; echo "rc $?" >> gpurun_out/r05b/pit_tests.log
tail -5 gpurun_out/r05b/pit_tests.log
FUZZ_SEED=401 timeout 200 python tools/probes/fuzz_point_in_tet.py 90 > gpurun_out/r05b/fuzz.log 2>&1; tail -3 gpurun_out/r05b/fuzz.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-brute-force --no-bandwidth-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('track', d['ms_per_step'], d.get('ms_per_step_median'), d['roofline']['avg_launch_ms'])" | tee -a gpurun_out/r05b/bench_ab.log
DEFTET_BENCH_QUERY_BOX=measure python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-brute-force --no-bandwidth-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('measure', d['ms_per_step'], d.get('ms_per_step_median'), d['roofline']['avg_launch_ms'])" | tee -a gpurun_out/r05b/bench_ab.log
done
