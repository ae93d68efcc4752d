cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  timeout 300 python -m pytest "tests/test_gpu_model.py::test_train_forward_backward_vs_oracle[s_psp-f32]" "tests/test_gpu_ops.py::test_block[rfb2-train-f32]" -m gpu -q --timeout 300 2>&1 | tail -1
done
echo "--- with MYOLO_NO_WGRAD_TILE / no side stream variations"
for E in "MYOLO_GRAPH_TRAIN=0" "MYOLO_NO_HALO=1 MYOLO_NO_WGRAD_TILE=1" "AMD_SERIALIZE_KERNEL=3"; do
  for i in 1 2 3; do
    echo -n "[$E] "; env $E timeout 300 python -m pytest "tests/test_gpu_model.py::test_train_forward_backward_vs_oracle[s_psp-f32]" "tests/test_gpu_model.py::test_train_forward_backward_vs_oracle[s_base-f32]" -m gpu -q --timeout 300 2>&1 | tail -1
  done
done
