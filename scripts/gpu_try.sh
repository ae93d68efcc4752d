cd $GRAFT_REPO_ROOT
for v in 0 1; do
  if [ $v = 1 ]; then export MYOLO_NO_WIDE_WGRAD=1; fi
  echo "no_wide=$v"; for i in 1 2; do timeout 300 python bench.py --steps 30 --warmup 5 --no-infer --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done
done
