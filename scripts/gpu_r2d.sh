cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "--- conv ubench"; timeout 900 python scripts/conv_ubench.py dbg > gpurun_out/r2d_ubench.log 2>&1; cat gpurun_out/r2d_ubench.log | tail -25 | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 600 -k "halo or stream" 2>&1 | tail -3
