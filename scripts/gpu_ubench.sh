# usage: bash scripts/gpu_ubench.sh <script.py> [args]  -- run one micro-benchmark on the GPU box, log to gpurun_out/ubench.log
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=${1:-bn_bench.py}; shift
timeout 300 python scripts/$S "$@" > gpurun_out/ubench.log 2>&1
tail -40 gpurun_out/ubench.log
