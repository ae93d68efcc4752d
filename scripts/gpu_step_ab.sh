# usage: bash scripts/gpu_step_ab.sh "<ENV=a> <ENV=b> ..." [rounds] -- in-step A/B of environment settings on ONE box, alternating, quick bench (no extras)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=${2:-2}
for r in $(seq $R); do
  for e in $1; do
    env ${e//,/ } timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-infer --no-kernel-timing --no-stock-baseline 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', round(d['ms_per_step'],3), round(d['value'],1))"
  done
done 2>&1 | tee gpurun_out/step_ab.txt
