# usage: bash scripts/gpu_prof_quick.sh <tag> [pytest -k expression]   -- optional parity subset, then rocprofv3 kernel stats of 5 bench steps + the timeline summary
TAG=${1:-q}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ -n "$2" ]; then timeout 600 python -m pytest tests -m gpu -q -x -k "$2" 2>&1 | tail -3; fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o train -- python bench.py --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1
tail -1 gpurun_out/prof_$TAG.log | cut -c1-200
python scripts/trace_timeline.py $(find gpurun_out/prof_$TAG -name 'train_kernel_trace.csv' | head -1) > gpurun_out/${TAG}_timeline.txt 2>&1; head -34 gpurun_out/${TAG}_timeline.txt
