cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python scripts/conv_ubench.py sdbg > gpurun_out/r2g_ubench.log 2>&1; grep "k1" gpurun_out/r2g_ubench.log | cut -c1-330
