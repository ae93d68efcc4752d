# usage: bash scripts/gpu_envs.sh "ENV1=a ENV2=b" "ENV1=c" ...   -- the training step of bench.py under each environment (same box), one line each
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for E in "$@"; do
  R=$(env $E timeout 300 python bench.py --steps ${STEPS:-30} --warmup 8 --no-cpu-baseline --no-infer 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('%.3f ms  %.0f img/s  conv %.3f ms (%d launches) frac %.3f' % (j['ms_per_step'], j['value'], r['avg_launch_us']*r['launches_per_step']/1e3, r['launches_per_step'], r['frac']))")
  echo "[$E]: $R" | tee -a gpurun_out/envs.txt
done
