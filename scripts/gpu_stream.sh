cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MYOLO_STREAM_MIN_TILES=1 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q --timeout 200 -k "f16" > gpurun_out/test_stream.log 2>&1; tail -12 gpurun_out/test_stream.log | cut -c1-300
echo "--- conv bench2 (stream)"; timeout 300 python scripts/conv_bench2.py 2>&1 | grep -v amdgpu
echo "--- conv bench2 (v1)"; MYOLO_NO_STREAM=1 timeout 300 python scripts/conv_bench2.py 2>&1 | grep -v amdgpu
echo "--- bench"; timeout 600 python bench.py --steps 20 --warmup 5 --no-infer --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
