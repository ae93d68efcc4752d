cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --durations=8 > gpurun_out/r2b_tests.log 2>&1
tail -40 gpurun_out/r2b_tests.log | cut -c1-400
echo "--- dbg sgd"; timeout 300 python scripts/dbg_sgd.py 2>&1 | grep -v Warning | tail -8
echo "--- bench graph seg16"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer --no-kernel-timing > gpurun_out/r2b_bench_seg.log 2>&1; tail -1 gpurun_out/r2b_bench_seg.log | cut -c1-300
echo "--- bench graph fork"; MYOLO_GRAPH_BWD=fork timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer --no-kernel-timing > gpurun_out/r2b_bench_fork.log 2>&1; tail -3 gpurun_out/r2b_bench_fork.log | cut -c1-300
