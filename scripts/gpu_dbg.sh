cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/dbg_staged2.py 2>&1 | tail -8 | cut -c1-600
