# usage: bash scripts/gpu_all.sh <tag> : full gpu test suite, smoke, bench, rocprof
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 200 > gpurun_out/test_$TAG.log 2>&1; tail -6 gpurun_out/test_$TAG.log | cut -c1-250
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
bash scripts/gpu_prof.sh $TAG
