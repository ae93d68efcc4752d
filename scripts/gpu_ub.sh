cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python scripts/wgrad_ubench.py 2>&1 | grep -v amdgpu.ids | cut -c1-220
