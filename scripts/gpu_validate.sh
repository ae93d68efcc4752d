# end-of-session validation on the GPU box: the whole GPU suite, smoke(), rocprofv3 kernel stats of the training step on this build
# (-> python scripts/prof_summary.py gpurun_out/prof_<tag>/train_kernel_stats.csv <tag> 7 <bench log> -> profiles/<tag>_summary.md).
# usage: bash scripts/gpu_validate.sh <tag> [bench]     ("bench": also the full bench line -> gpurun_out/bench_<tag>.log)
TAG=${1:-val}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "--- suite"; timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/suite_$TAG.log 2>&1; tail -4 gpurun_out/suite_$TAG.log | cut -c1-300
echo "--- smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-300
if [ "$2" = "bench" ]; then echo "--- bench"; timeout 600 python bench.py > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log | cut -c1-3000; fi
echo "--- prof"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o train -- python bench.py --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1; tail -1 gpurun_out/prof_$TAG.log | cut -c1-300
echo "--- train.py loop, pruned one-loss backward lists on / off"; timeout 300 python scripts/train_py_ab.py 2>&1 | tail -1 | cut -c1-800
