cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_optim.py tests/test_gpu_loss.py -m gpu -q --timeout 120 > gpurun_out/t5.log 2>&1; tail -15 gpurun_out/t5.log | cut -c1-200
timeout 600 python bench.py --steps 10 --warmup 3 --no-infer > gpurun_out/b5.log 2>&1
tail -5 gpurun_out/b5.log
