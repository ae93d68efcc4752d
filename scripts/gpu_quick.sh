cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_upce.py tests/test_gpu_postproc.py tests/test_gpu_model.py -m gpu -q -x --timeout 600 2>&1 | tail -6 | cut -c1-300
echo "--- host time"; timeout 300 python scripts/host_time.py 2>&1 | grep -E "host enqueue|native forward|run_backward|engine.py" | cut -c1-200
echo "--- nms"; timeout 300 python scripts/nms_bench.py detect 2>&1 | tail -2
timeout 300 python bench.py --stage infer --infer-size 1024 2048 --steps 60 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-420
timeout 300 python bench.py --stage infer --infer-size 512 1024 --steps 60 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-420
timeout 600 python bench.py --steps 30 --warmup 8 --no-infer --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
