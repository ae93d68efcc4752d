# usage: bash scripts/gpu_round.sh <tag> : full GPU parity suite, host time of a step, NMS timing, inference bench + kernel trace at both sizes
TAG=${1:-r3b}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -8 | cut -c1-400
echo "--- host time"; timeout 300 python scripts/host_time.py 2>&1 | head -3
echo "--- nms"; timeout 300 python scripts/nms_bench.py 2>&1 | tail -4
bash scripts/gpu_inf.sh $TAG 2>&1 | cut -c1-330
