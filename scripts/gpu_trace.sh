# usage: bash scripts/gpu_trace.sh <tag> [ENV=..]  -- rocprofv3 kernel trace (timestamps) of 5 bench steps -> gpurun_out/trace_<tag>/
TAG=${1:-t}; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
env "$@" timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$TAG -o tr -- python bench.py --steps 4 --warmup 3 --no-kernel-timing --no-infer --no-cpu-baseline > gpurun_out/trace_$TAG.log 2>&1
tail -1 gpurun_out/trace_$TAG.log | cut -c1-200
find gpurun_out/trace_$TAG -name "*.csv" | head
python scripts/trace_timeline.py $(find gpurun_out/trace_$TAG -name "*kernel_trace.csv" | head -1) | tail -60
