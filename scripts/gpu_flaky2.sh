cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T="tests/test_gpu_model.py::test_train_forward_backward_vs_oracle tests/test_gpu_ops.py::test_block"
run() { echo -n "[$1] "; shift; timeout 600 python -m pytest "$@" -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -1; }
run "dropin-ddp + model/ops" "tests/test_gpu_dropin.py::test_train_py_loop_body_with_stock_sgd_amp_ddp_then_checkpoint_roundtrip" $T
run "dropin-sgd + model/ops" "tests/test_gpu_dropin.py::test_stock_sgd_step_equals_fused_sgd_step" $T
run "dropout + model/ops" tests/test_gpu_dropout.py $T
run "loss + model/ops" tests/test_gpu_loss.py $T
run "amp steps + model/ops" "tests/test_gpu_model.py::test_amp_training_steps_run_and_learn" $T
MYOLO_GRAPH_TRAIN=0 run "GRAPH_TRAIN=0 dropin(all) + model/ops" tests/test_gpu_dropin.py $T
