cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q --maxfail=12 --timeout 900 2>&1 | tail -40 | cut -c1-400
echo "--- host time"; timeout 300 python scripts/host_time.py 2>&1 | grep -v "^$" | head -40 | cut -c1-200
