cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "--- wgrad ubench"; timeout 900 python scripts/wgrad_ubench.py > gpurun_out/r2e_wgrad.log 2>&1; tail -22 gpurun_out/r2e_wgrad.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 600 -x 2>&1 | tail -5
echo "--- bench eager"; MYOLO_GRAPH_TRAIN=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer --no-kernel-timing > gpurun_out/r2e_bench_eager.log 2>&1; tail -1 gpurun_out/r2e_bench_eager.log | cut -c1-300
echo "--- bench graph"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer --no-kernel-timing > gpurun_out/r2e_bench_graph.log 2>&1; tail -1 gpurun_out/r2e_bench_graph.log | cut -c1-300
