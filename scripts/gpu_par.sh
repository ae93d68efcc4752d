# quick look at the experimental branch-parallel forward (MYOLO_PAR=1): one full-resolution parity test + the detect.py FPS
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MYOLO_PAR=1 timeout 120 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "full_resolution" > gpurun_out/par.log 2>&1; tail -3 gpurun_out/par.log | cut -c1-300
echo -n "PAR=1: "; MYOLO_PAR=1 timeout 120 python bench.py --stage infer 2>&1 | tail -1 | cut -c1-160
