# usage: bash scripts/gpu_sweep.sh "ENV1=a ENV2=b" "ENV1=c" ...   -- bench.py (eager + graph) under each environment, one line each
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for E in "$@"; do
  for G in ${SWEEP_GRAPHS:-0 1}; do
    R=$(env $E MYOLO_GRAPH_TRAIN=$G timeout 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-infer --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.3f ms  %.0f img/s' % (j['ms_per_step'], j['value']))")
    echo "[$E] graph=$G: $R"
  done
done
