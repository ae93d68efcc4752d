# round-2 first GPU call: full parity suite + bench with / without plan-level hipGraphs
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -x --durations=8 > gpurun_out/r2a_tests.log 2>&1
tail -25 gpurun_out/r2a_tests.log | cut -c1-600
echo "--- bench graph"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2a_bench_graph.log 2>&1; tail -2 gpurun_out/r2a_bench_graph.log | cut -c1-900
echo "--- bench eager"; MYOLO_GRAPH_TRAIN=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer > gpurun_out/r2a_bench_eager.log 2>&1; tail -1 gpurun_out/r2a_bench_eager.log | cut -c1-400
