cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 600 2>&1 | tail -3
timeout 600 python scripts/wgrad_ubench.py 2>&1 | grep "k1" | cut -c1-200
bash scripts/gpu_sweep.sh "MYOLO_X=1"
