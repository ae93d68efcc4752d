cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q --timeout 600 2>&1 | tail -4
bash scripts/gpu_sweep.sh "MYOLO_X=1" "MYOLO_NO_HALO_S2=1"
