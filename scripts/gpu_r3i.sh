# one-call validation of the round-3 second session: new parity tests first, then the whole GPU suite, the bench line, the
# train.py-loop A/B.  usage: bash scripts/gpu_r3i.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "--- new tests"; timeout 400 python -m pytest tests/test_gpu_prune.py -m gpu -q --timeout 300 > gpurun_out/prune_tests.log 2>&1; tail -25 gpurun_out/prune_tests.log | cut -c1-300
echo "--- suite"; timeout 900 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_gpu_prune.py > gpurun_out/suite.log 2>&1; tail -8 gpurun_out/suite.log | cut -c1-300
echo "--- bench"; timeout 600 python bench.py > gpurun_out/bench_r3i.log 2>&1; tail -1 gpurun_out/bench_r3i.log | cut -c1-3000
echo "--- train.py loop A/B"; timeout 300 python scripts/train_py_ab.py 2>&1 | tail -1 | cut -c1-800
