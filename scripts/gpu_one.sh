# run a pytest selection on the GPU box with full tracebacks: bash scripts/gpu_one.sh "<pytest args>"
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
eval timeout 600 python -m pytest "$1" -m gpu -q --timeout 300 -x --durations=5 > gpurun_out/one.log 2>&1
tail -60 gpurun_out/one.log | cut -c1-400
