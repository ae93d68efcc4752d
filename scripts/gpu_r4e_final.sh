# end-of-session validation, round 4 second session: whole GPU suite, smoke, the bench line, rocprofv3 kernel stats + per-layer conv table + timeline,
# the two PMC passes of the same command.  usage: bash scripts/gpu_r4e_final.sh <tag>
TAG=${1:-r4e}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "--- suite"; timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/suite_$TAG.log 2>&1; tail -4 gpurun_out/suite_$TAG.log | cut -c1-300
echo "--- smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-400
echo "--- bench"; timeout 900 python bench.py --no-stock-baseline > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log | cut -c1-1500
echo "--- prof"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o train -- python bench.py --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/prof_$TAG -name 'train_kernel_stats.csv' | head -1) $TAG 7 gpurun_out/bench_$TAG.log
python scripts/conv_trace.py $(find gpurun_out/prof_$TAG -name 'train_kernel_trace.csv' | head -1) > gpurun_out/${TAG}_conv_layers.txt 2>&1; head -4 gpurun_out/${TAG}_conv_layers.txt
python scripts/trace_timeline.py $(find gpurun_out/prof_$TAG -name 'train_kernel_trace.csv' | head -1) > gpurun_out/${TAG}_timeline.txt 2>&1; head -4 gpurun_out/${TAG}_timeline.txt
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$C -o x -- python bench.py --steps 2 --warmup 1 --no-kernel-timing --no-infer --no-cpu-baseline > gpurun_out/pmc_${TAG}_$C.log 2>&1
  tail -1 gpurun_out/pmc_${TAG}_$C.log | cut -c1-120
done
python scripts/pmc_summary.py $TAG 3 2>&1 | tail -8
cp profiles/${TAG}_* gpurun_out/ 2>/dev/null
rm -rf gpurun_out/prof_$TAG/*/*.db 2>/dev/null
du -sh gpurun_out | tail -1
