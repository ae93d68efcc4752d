cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python scripts/conv_small_ubench.py 512 1024 2>&1 | grep -v amdgpu.ids | cut -c1-200
timeout 900 python scripts/conv_small_ubench.py 1024 2048 2>&1 | grep -v amdgpu.ids | cut -c1-200
