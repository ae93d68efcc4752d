# second validation call of the session: SyncBatchNorm (2 ranks on one GPU) + the BatchNorm / block parity tests on the rebuilt kernels,
# then a kernel trace of the training step for the per-queue timeline.  usage: bash scripts/gpu_r3j.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "--- dist tests"; timeout 500 python -m pytest tests/test_gpu_dist.py -m gpu -q --timeout 400 > gpurun_out/dist_tests.log 2>&1; tail -30 gpurun_out/dist_tests.log | cut -c1-400
echo "--- ops + model tests"; timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_prune.py -m gpu -q --timeout 400 -x > gpurun_out/ops_tests.log 2>&1; tail -6 gpurun_out/ops_tests.log | cut -c1-300
echo "--- trace"; bash scripts/gpu_trace.sh r3j 2>&1 | tail -75 | cut -c1-200
