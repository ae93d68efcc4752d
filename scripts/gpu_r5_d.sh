# round 5, call D: stale-threshold sweep of the training step (every switch below was last measured before conv_mid / conv_midx existed or
# on an older build); one box, baseline first / middle / last
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "--- dense stride-2 dgrad: tests"; timeout 900 python -m pytest tests/test_gpu_s2_dense.py tests/test_gpu_bench_plan.py "tests/test_gpu_configs.py::test_fp16_conv_layer_at_its_real_shape" "tests/test_gpu_configs.py::test_fp16_block_at_its_real_shape" "tests/test_gpu_model.py::test_full_resolution_joint_train_step_vs_oracle" -m gpu -q --timeout 600 > gpurun_out/r5d_tests.log 2>&1; tail -15 gpurun_out/r5d_tests.log | cut -c1-300
echo "--- pair ubench"; timeout 300 python scripts/pair_ubench.py 2>&1 | tail -10 | tee gpurun_out/r5d_pair_ubench.txt
for E in "A=0" "MYOLO_S2_DENSE=0" "MYOLO_S2_DENSE=0 MYOLO_NO_HALO_S2=1" "MYOLO_BN_STATS_MAX_ELEMS=9000000" "MYOLO_BN_STATS_MAX_ELEMS=17000000" "MYOLO_BN_APPLY_FOLD_MAXK=128" "MYOLO_BN_APPLY_FOLD_MAXK=512" "A=1" \
         "MYOLO_BN_WGS_FWD=512" "MYOLO_BN_WGS_APPLY=512" "MYOLO_BN_WGS_REDUCE=256" "MYOLO_BN_WGS_REDUCE=1024" "MYOLO_WGRAD_TILE_WG=96" "MYOLO_WGRAD_TILE_WG=160" "MYOLO_CONV_MIDX=0" "MYOLO_STREAM_MIN_TILES=100000" "MYOLO_NO_HALO=1" "A=2" $EXTRA; do
  R=$(env $E timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-infer --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.3f ms  %.0f img/s' % (j['ms_per_step'], j['value']))" 2>&1 | tail -1)
  echo "[$E] train: $R" | tee -a gpurun_out/r5d_sweep.txt
done
