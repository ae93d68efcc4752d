# round 5, call C: whole suite on the new defaults (tiny conv on, fused Bottleneck pairs in eval plans), FPS with the pair kernel off / on,
# kernel trace + per-frame timeline of the detect.py loop on the new build
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/bench_plan_variants.txt
echo "--- whole suite"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --durations=8 > gpurun_out/r5c_suite.log 2>&1; tail -25 gpurun_out/r5c_suite.log | cut -c1-300
echo "--- detect.py frame, pair off / on"
for E in "MYOLO_CONV_PAIR=0" "MYOLO_CONV_PAIR=1" "MYOLO_CONV_PAIR=0" "MYOLO_CONV_PAIR=1"; do
  for SZ in "1024 2048" "512 1024"; do
    R=$(env $E timeout 300 python bench.py --stage infer --infer-size $SZ --steps 300 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.0f FPS  launches %s stages %s' % (j['value'], j.get('forward_launches'), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in j.get('stage_ms', {}).items() if k != 'what'}))" 2>&1 | tail -1)
    echo "[$E] infer $SZ: $R" | tee -a gpurun_out/r5c_pair.txt
  done
done
echo "--- frame timeline"
for S in "1024 2048" "512 1024"; do
  T=$(echo $S | tr ' ' 'x')
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5c_itrace_$T -o tr -- python bench.py --stage infer --infer-size $S --steps 60 --no-cpu-baseline > gpurun_out/r5c_itrace_$T.log 2>&1
  python scripts/trace_infer_timeline.py $(find gpurun_out/r5c_itrace_$T -name "*kernel_trace.csv" | head -1) > gpurun_out/r5c_infer_timeline_$T.txt 2>&1
  head -14 gpurun_out/r5c_infer_timeline_$T.txt | cut -c1-200
  cp $(find gpurun_out/r5c_itrace_$T -name "*kernel_stats.csv" | head -1) gpurun_out/r5c_infer${T}_kernel_stats.csv 2>/dev/null
  rm -rf gpurun_out/r5c_itrace_$T
done
