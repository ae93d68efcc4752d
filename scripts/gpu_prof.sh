# usage: bash scripts/gpu_prof.sh <tag>   -- bench + rocprofv3 kernel trace of the training step (+ inference) on the GPU box
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.log 2>&1
tail -1 gpurun_out/bench_$TAG.log | cut -c1-1500
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o train -- python bench.py --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1
tail -2 gpurun_out/prof_$TAG.log | cut -c1-300
ls gpurun_out/prof_$TAG | head
