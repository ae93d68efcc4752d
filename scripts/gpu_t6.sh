cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_postproc.py -m gpu -q --timeout 200 > gpurun_out/t6.log 2>&1; tail -40 gpurun_out/t6.log | cut -c1-220
timeout 300 python bench.py --stage infer > gpurun_out/b6.log 2>&1; tail -3 gpurun_out/b6.log | cut -c1-400
