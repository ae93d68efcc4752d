cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "unjoined" 2>&1 | tail -12
