cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/wgrad_ubench.py 2>&1 | tail -17 | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 600 2>&1 | tail -2
bash scripts/gpu_sweep.sh "MYOLO_X=1" "MYOLO_WGRAD_WG_HINT=192" "MYOLO_NO_SIDE=1"
