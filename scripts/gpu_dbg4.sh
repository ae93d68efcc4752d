cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 200 python scripts/bilbwd_bench.py 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 300 -x -k "pyramid or bilinear" 2>&1 | tail -3 | cut -c1-300
