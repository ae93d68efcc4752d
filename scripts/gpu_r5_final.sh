# end-of-round validation, round 5: whole GPU suite (fresh parity log), smoke, the bench line, rocprofv3 kernel stats + per-layer conv table + timeline of the
# training step, the two PMC passes of the same command, kernel stats + timeline of the detect.py frame.  usage: bash scripts/gpu_r5_final.sh <tag>
TAG=${1:-r5}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_log.jsonl gpurun_out/bench_plan_variants.txt
echo "--- suite"; timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/suite_$TAG.log 2>&1; tail -4 gpurun_out/suite_$TAG.log | cut -c1-300
python scripts/parity_summary.py > gpurun_out/${TAG}_parity_summary.md 2>&1; head -3 gpurun_out/${TAG}_parity_summary.md | cut -c1-200
echo "--- smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-400
echo "--- bench"; timeout 900 python bench.py > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log | cut -c1-1200
echo "--- prof"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o train -- python bench.py --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/prof_$TAG -name 'train_kernel_stats.csv' | head -1) $TAG 7 gpurun_out/bench_$TAG.log
python scripts/conv_trace.py $(find gpurun_out/prof_$TAG -name 'train_kernel_trace.csv' | head -1) > gpurun_out/${TAG}_conv_layers.txt 2>&1; head -4 gpurun_out/${TAG}_conv_layers.txt
python scripts/trace_timeline.py $(find gpurun_out/prof_$TAG -name 'train_kernel_trace.csv' | head -1) > gpurun_out/${TAG}_timeline.txt 2>&1; head -4 gpurun_out/${TAG}_timeline.txt
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$C -o x -- python bench.py --steps 2 --warmup 1 --no-kernel-timing --no-infer --no-cpu-baseline > gpurun_out/pmc_${TAG}_$C.log 2>&1
  tail -1 gpurun_out/pmc_${TAG}_$C.log | cut -c1-120
done
python scripts/pmc_summary.py $TAG 3 2>&1 | tail -8
echo "--- frame"
for S in "1024 2048" "512 1024"; do
  T=$(echo $S | tr ' ' 'x')
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_itrace_$T -o tr -- python bench.py --stage infer --infer-size $S --steps 60 --no-cpu-baseline > gpurun_out/${TAG}_itrace_$T.log 2>&1
  python scripts/trace_infer_timeline.py $(find gpurun_out/${TAG}_itrace_$T -name "*kernel_trace.csv" | head -1) > gpurun_out/${TAG}_infer_timeline_$T.txt 2>&1
  head -4 gpurun_out/${TAG}_infer_timeline_$T.txt | cut -c1-200
  cp $(find gpurun_out/${TAG}_itrace_$T -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_infer${T}_kernel_stats.csv 2>/dev/null
  rm -rf gpurun_out/${TAG}_itrace_$T
done
cp profiles/${TAG}_* gpurun_out/ 2>/dev/null
rm -rf gpurun_out/prof_$TAG/*/*.db 2>/dev/null
du -sh gpurun_out | tail -1
