# round 5, call B: the fused Bottleneck kernel (conv_pair.hip): C-ABI parity, the frame plans launch by launch, FPS with it off / on;
# the tiny-conv default (on / off step + the on/off gradient comparison), the three re-toleranced two-run tests
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/bench_plan_variants.txt
echo "--- tests"; timeout 1500 python -m pytest tests/test_gpu_conv_pair.py tests/test_gpu_bench_plan.py tests/test_gpu_tiny_conv.py tests/test_gpu_dist.py tests/test_gpu_dropin.py "tests/test_gpu_prune.py::test_pruned_backward_equals_the_full_list_in_any_order" -m gpu -q --timeout 600 --durations=6 > gpurun_out/r5b_tests.log 2>&1; tail -40 gpurun_out/r5b_tests.log | cut -c1-300
echo "--- detect.py frame, pair off / on"
for E in "MYOLO_CONV_PAIR=0" "MYOLO_CONV_PAIR=1" "MYOLO_CONV_PAIR=0" "MYOLO_CONV_PAIR=1"; do
  for SZ in "1024 2048" "512 1024"; do
    R=$(env $E timeout 300 python bench.py --stage infer --infer-size $SZ --steps 200 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.0f FPS  launches %s stages %s' % (j['value'], j.get('forward_launches'), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in j.get('stage_ms', {}).items() if k != 'what'}))" 2>&1 | tail -1)
    echo "[$E] infer $SZ: $R" | tee -a gpurun_out/r5b_pair.txt
  done
done
echo "--- train step, tiny off / on"
for E in "MYOLO_TINY_CONV=0" "MYOLO_TINY_CONV=1" "MYOLO_TINY_CONV=0" "MYOLO_TINY_CONV=1"; do
  R=$(env $E timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-infer --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.3f ms  %.0f img/s' % (j['ms_per_step'], j['value']), j['checks'])" 2>&1 | tail -1)
  echo "[$E] train: $R" | tee -a gpurun_out/r5b_tiny.txt
done
