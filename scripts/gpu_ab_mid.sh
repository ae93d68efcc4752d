# usage: bash scripts/gpu_ab_mid.sh  -- full GPU suite, then the training step with conv_mid off / on (same box)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 -x 2>&1 | tail -8 | cut -c1-400 | tee gpurun_out/ab_mid_tests.txt
for M in 0 1 2; do
  R=$(MYOLO_CONV_MID=$M timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-infer 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('%.3f ms  %.0f img/s  conv %.3f ms (%d launches) frac %.3f' % (j['ms_per_step'], j['value'], r['avg_launch_us']*r['launches_per_step']/1e3, r['launches_per_step'], r['frac']))")
  echo "MYOLO_CONV_MID=$M: $R" | tee -a gpurun_out/ab_mid.txt
done
