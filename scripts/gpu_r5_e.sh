# round 5, call E: gated pair kernel + 192-wide conv_mid tile + 4-deep adaptive-pool loads: tests, FPS, second threshold sweep, yolov5m + Lab
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/bench_plan_variants.txt
echo "--- tests"; timeout 1200 python -m pytest tests/test_gpu_conv_pair.py tests/test_gpu_conv_mid.py tests/test_gpu_bench_plan.py tests/test_gpu_ops.py "tests/test_gpu_configs.py::test_fp16_block_at_its_real_shape" -m gpu -q --timeout 600 > gpurun_out/r5e_tests.log 2>&1; tail -12 gpurun_out/r5e_tests.log | cut -c1-300
echo "--- detect.py frame"
for E in "MYOLO_CONV_PAIR=0" "A=1" "MYOLO_CONV_PAIR=0" "A=1"; do
  for SZ in "1024 2048" "512 1024"; do
    R=$(env $E timeout 300 python bench.py --stage infer --infer-size $SZ --steps 300 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.0f FPS  launches %s stages %s' % (j['value'], j.get('forward_launches'), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in j.get('stage_ms', {}).items() if k != 'what'}))" 2>&1 | tail -1)
    echo "[$E] infer $SZ: $R" | tee -a gpurun_out/r5e_pair.txt
  done
done
echo "--- yolov5m + Lab bs 8"
for E in "MYOLO_MID_NO_BN192=1" "A=1" "MYOLO_MID_NO_BN192=1" "A=1"; do
  R=$(env $E timeout 300 python bench.py --cfg yolov5m_city_seg_lab.yaml --batch 8 --steps 30 --warmup 8 --no-cpu-baseline --no-infer --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.3f ms  %.0f img/s' % (j['ms_per_step'], j['value']))" 2>&1 | tail -1)
  echo "[$E] m+lab: $R" | tee -a gpurun_out/r5e_mlab.txt
done
echo "--- train sweep 2"
for E in "A=0" "MYOLO_BN_WGS_APPLY=256" "MYOLO_BN_WGS_APPLY=384" "MYOLO_BN_WGS_APPLY=512" "MYOLO_BN_WGS_APPLY=768" "A=1" "MYOLO_BN_SLICE=32" "MYOLO_BN_SLICE=128" "MYOLO_WGRAD_TILE_MIN_TILES=4" "MYOLO_WGRAD_TILE_MIN_TILES=8" "MYOLO_OPTIM_CHUNK=4096" "MYOLO_OPTIM_CHUNK=16384" "MYOLO_BN_WGS_APPLY=512 MYOLO_BN_APPLY_FOLD_MAXK=128" "A=2"; do
  R=$(env $E timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-infer --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.3f ms  %.0f img/s' % (j['ms_per_step'], j['value']))" 2>&1 | tail -1)
  echo "[$E] train: $R" | tee -a gpurun_out/r5e_sweep.txt
done
