# XCD-aware workgroup order of wgrad_tile: per-layer A/B + numerics, real-shape tests, the step with / without
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python scripts/wgrad_ubench.py xcd > gpurun_out/wxcd_ubench.txt 2>&1; tail -19 gpurun_out/wxcd_ubench.txt | cut -c1-250
timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py -q -x -k "(real_shape and train) or full_resolution_joint_train_step_vs_oracle" 2>&1 | tail -3 | cut -c1-300
for E in "MYOLO_WGRAD_TILE_XCD=0" "MYOLO_WGRAD_TILE_XCD=1" "MYOLO_WGRAD_TILE_XCD=0" "MYOLO_WGRAD_TILE_XCD=1" "MYOLO_WGRAD_TILE_XCD=1 MYOLO_WGRAD_TILE_DMA=0"; do
  R=$(env $E timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-infer --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.3f ms  %.0f img/s' % (j['ms_per_step'], j['value']))")
  echo "[$E]: $R" | tee -a gpurun_out/wxcd_step.txt
done
