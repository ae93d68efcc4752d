cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/host_time_infer.py 2>&1 | grep -v amdgpu | cut -c1-160 | head -48
