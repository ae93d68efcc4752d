# round 5, call A: (1) the new per-launch tests of the benchmarked plans + the pruned weight-gradient variants, (2) the WHOLE suite with
# MYOLO_TINY_CONV=1 (decides its default), (3) which hardware queue the second stream should sit on (detect.py frame + training step)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/bench_plan_variants.txt
echo "--- new tests"; timeout 900 python -m pytest tests/test_gpu_bench_plan.py tests/test_gpu_wgrad_tile.py -m gpu -q --timeout 600 --durations=6 > gpurun_out/r5a_new.log 2>&1; tail -40 gpurun_out/r5a_new.log | cut -c1-300
echo "--- whole suite, MYOLO_TINY_CONV=1"; MYOLO_TINY_CONV=1 timeout 1200 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_gpu_bench_plan.py > gpurun_out/r5a_tiny_suite.log 2>&1; tail -30 gpurun_out/r5a_tiny_suite.log | cut -c1-300
echo "--- second-stream placement"
for E in "A=0" "MYOLO_SIDE_SKIP=1" "MYOLO_SIDE_SKIP=2" "MYOLO_SIDE_SKIP=3" "MYOLO_SIDE_PRIO=-1" "GPU_MAX_HW_QUEUES=8" "A=1"; do
  for SZ in "1024 2048"; do
    R=$(env $E timeout 300 python bench.py --stage infer --infer-size $SZ --steps 200 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.0f FPS  stages %s' % (j['value'], {k: (round(v, 3) if isinstance(v, float) else v) for k, v in j.get('stage_ms', {}).items() if k != 'what'}))" 2>&1 | tail -1)
    echo "[$E] infer $SZ: $R" | tee -a gpurun_out/r5a_side.txt
  done
  R=$(env $E timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-infer --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.3f ms  %.0f img/s' % (j['ms_per_step'], j['value']))" 2>&1 | tail -1)
  echo "[$E] train: $R" | tee -a gpurun_out/r5a_side.txt
done
