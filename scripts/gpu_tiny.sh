# csrc/tiny_conv.hip: C-ABI parity, the block / model / pruned-backward tests with the tiny-conv groups ON, the step with and without
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "--- C ABI"; timeout 300 python -m pytest tests/test_gpu_tiny_conv.py -q 2>&1 | tail -25 | cut -c1-400
echo "--- suite subset, MYOLO_TINY_CONV=1"; MYOLO_TINY_CONV=1 timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_prune.py -q -k "test_block or train_forward_backward_vs_oracle or prune or amp_training or (full_resolution_joint_train_step_vs_oracle and f16)" 2>&1 | tail -25 | cut -c1-400
for E in "MYOLO_TINY_CONV=0" "MYOLO_TINY_CONV=1" "MYOLO_TINY_CONV=0" "MYOLO_TINY_CONV=1"; do
  R=$(env $E timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-infer --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.3f ms  %.0f img/s' % (j['ms_per_step'], j['value']), j['checks'])" 2>&1 | tail -1)
  echo "[$E]: $R" | tee -a gpurun_out/tiny_step.txt
done
