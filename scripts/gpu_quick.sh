cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_upce.py -x -q -k eval_logits 2>&1 | grep -v Warning | tail -30
