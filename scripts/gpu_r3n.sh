# final validation of the second session of round 3: the whole GPU suite, smoke(), rocprofv3 kernel stats of the training step on
# this build (-> scripts/prof_summary.py -> profiles/r3i_summary.md).  usage: bash scripts/gpu_r3n.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "--- suite"; timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/suite_final.log 2>&1; tail -4 gpurun_out/suite_final.log | cut -c1-300
echo "--- smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-300
echo "--- prof"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3i -o train -- python bench.py --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline > gpurun_out/prof_r3i.log 2>&1; tail -1 gpurun_out/prof_r3i.log | cut -c1-300
find gpurun_out/prof_r3i -name '*.csv' | head
