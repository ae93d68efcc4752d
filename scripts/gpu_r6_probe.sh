# round 6 probe: new C-ABI tests, yolov5m ragged-K A/B, kernel trace of the step with and without the weight-gradient stream
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "--- tests"; timeout 1500 python -m pytest tests/test_gpu_conv_pair.py tests/test_gpu_conv_mid.py tests/test_gpu_bench_plan.py tests/test_gpu_bn_fused.py -q 2>&1 | tail -15
echo "--- mlab A/B"
for r in 1 2; do for e in 0 1; do
  MYOLO_MID_RAGGED=$e timeout 300 python bench.py --cfg yolov5m_city_seg_lab.yaml --batch 8 --steps 30 --warmup 8 --no-cpu-baseline --no-infer --no-kernel-timing --no-stock-baseline 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MID_RAGGED=$e', round(d['ms_per_step'],3), round(d['value'],1))"
done; done 2>&1 | tee gpurun_out/mlab_ragged_ab.txt
echo "--- trace"
for T in base nowgrad; do
  E=""; [ $T = nowgrad ] && E="MYOLO_DBG_SKIP_WGRAD=1 MYOLO_NATIVE_EXEC=0"
  env $E timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$T -o train -- python bench.py --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline --no-stock-baseline > gpurun_out/prof_$T.log 2>&1
  tail -1 gpurun_out/prof_$T.log | cut -c1-160
  TR=$(find gpurun_out/prof_$T -name 'train_kernel_trace.csv' | head -1)
  python scripts/conv_trace.py $TR > gpurun_out/r6p_conv_layers_$T.txt 2>&1; head -4 gpurun_out/r6p_conv_layers_$T.txt
  python scripts/trace_timeline.py $TR > gpurun_out/r6p_timeline_$T.txt 2>&1; head -4 gpurun_out/r6p_timeline_$T.txt
  python scripts/trace_list.py $TR > gpurun_out/r6p_list_$T.txt 2>&1
  cp $(find gpurun_out/prof_$T -name 'train_kernel_stats.csv' | head -1) gpurun_out/r6p_kernel_stats_$T.csv
  rm -rf gpurun_out/prof_$T
done
