cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r2k_tests.log 2>&1
tail -5 gpurun_out/r2k_tests.log | cut -c1-300
bash scripts/gpu_sweep.sh "MYOLO_X=1"
