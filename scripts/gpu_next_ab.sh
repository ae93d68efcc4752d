# opt-in switches prepared at the end of round 4 (no GPU minutes were left to run them): numerics first, then the step under each, same box.
#   MYOLO_WGRAD_TILE_WS1X1=1  1x1 weight gradients: split-K partials through the workspace + reduce launch instead of 128 x 16 K fp32 atomics
#   MYOLO_SIDE_BATCH=K        native executor: fork the weight-gradient stream once per K launches (79 event records per step today)
#   MYOLO_TINY_CONV=1         one-workgroup Conv+BatchNorm layers (scripts/gpu_tiny_flip.sh decides its default)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "--- ws1x1: per-layer A/B + numerics"; timeout 300 python scripts/wgrad_ubench.py ws1 > gpurun_out/next_ws1_ubench.txt 2>&1; tail -19 gpurun_out/next_ws1_ubench.txt | cut -c1-220
echo "--- joint step vs oracle under the switches"; for E in "MYOLO_WGRAD_TILE_WS1X1=1" "MYOLO_SIDE_BATCH=4"; do
  env $E timeout 300 python -m pytest tests/test_gpu_model.py -q -x -k "full_resolution_joint_train_step_vs_oracle and f16" 2>&1 | tail -2 | cut -c1-200
done
for E in "A=0" "MYOLO_SIDE_BATCH=2" "MYOLO_SIDE_BATCH=4" "MYOLO_SIDE_BATCH=8" "MYOLO_WGRAD_TILE_WS1X1=1" "MYOLO_SIDE_BATCH=4 MYOLO_WGRAD_TILE_WS1X1=1 MYOLO_TINY_CONV=1" "A=1"; do
  R=$(env $E timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-infer --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.3f ms  %.0f img/s' % (j['ms_per_step'], j['value']), j['checks'])" 2>&1 | tail -1)
  echo "[$E]: $R" | tee -a gpurun_out/next_ab_step.txt
done
