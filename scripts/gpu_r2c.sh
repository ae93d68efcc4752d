cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --durations=5 > gpurun_out/r2c_tests.log 2>&1
tail -30 gpurun_out/r2c_tests.log | cut -c1-300
echo "--- conv ubench"; timeout 600 python scripts/conv_ubench.py > gpurun_out/r2c_ubench.log 2>&1; cat gpurun_out/r2c_ubench.log | tail -25
echo "--- bench eager"; MYOLO_GRAPH_TRAIN=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer > gpurun_out/r2c_bench_eager.log 2>&1; tail -1 gpurun_out/r2c_bench_eager.log | cut -c1-1200
echo "--- bench graph"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer --no-kernel-timing > gpurun_out/r2c_bench_graph.log 2>&1; tail -1 gpurun_out/r2c_bench_graph.log | cut -c1-300
