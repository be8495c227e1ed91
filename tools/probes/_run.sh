cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_a.json 2> gpurun_out/r3_bench_a.err; tail -c 3000 gpurun_out/r3_bench_a.json
