cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/segup_bench.py 2>&1 | tail -5
