# wgrad_tile beside the chain: LDS budget / workgroup count of the LDS-DMA kernel (a smaller footprint leaves room for a conv_mid workgroup on the same CU)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for E in "$@"; do
  R=$(env $E timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-infer --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.3f ms  %.0f img/s' % (j['ms_per_step'], j['value']))")
  echo "[$E]: $R" | tee -a gpurun_out/wgrad_lds.txt
done
