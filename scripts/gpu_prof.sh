cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --stage fwdbwd --steps 10 --warmup 3 > gpurun_out/b3.log 2>&1
tail -5 gpurun_out/b3.log
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof3 -o fwdbwd -- python bench.py --stage fwdbwd --steps 5 --warmup 2 --no-kernel-timing > gpurun_out/p3.log 2>&1
tail -3 gpurun_out/p3.log
ls -R gpurun_out/prof3 | head
