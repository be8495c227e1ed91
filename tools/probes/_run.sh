cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_surface_ops_gpu.py tests/test_pipeline_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -3
python tools/step_demo.py --surface 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
python tools/probes/surface_batched_probe.py 2>&1 | tail -1
