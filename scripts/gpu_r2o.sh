cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 600 -k "bn_backward_sums" 2>&1 | tail -15 | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q --timeout 600 2>&1 | tail -5 | cut -c1-250
bash scripts/gpu_sweep.sh "MYOLO_X=1" "MYOLO_BN_STATS_IN_DGRAD=0"
