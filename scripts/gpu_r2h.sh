cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python scripts/conv_ubench.py > gpurun_out/r2h_ubench.log 2>&1; tail -17 gpurun_out/r2h_ubench.log | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q --timeout 600 2>&1 | tail -4
bash scripts/gpu_sweep.sh "MYOLO_WGRAD_TILE_WG=128" "MYOLO_WGRAD_TILE_WG=192"
