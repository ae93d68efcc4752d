cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for S in "$@"; do
  R=$(MYOLO_MID_SKIP=$S timeout 300 python -m pytest tests/test_gpu_model.py -k "test_full_resolution_joint_train_step_vs_oracle and f16" -q -x 2>&1 | grep -E "parameter gradients off|passed|failed" | tail -2 | tr '\n' ' ')
  echo "SKIP=$S: $R" | tee -a gpurun_out/bisect.txt
done
