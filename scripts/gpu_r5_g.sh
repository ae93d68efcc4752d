# round 5, call G: kernel-selection thresholds of the detect.py frame against the round-5 build, and the weight-gradient tail hints of the training step
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for E in "A=0" "MYOLO_SMALL_MAX_TILES=64" "MYOLO_SMALL_MAX_TILES=192" "MYOLO_SMALL_MAX_TILES=256" "MYOLO_NO_SMALL=1" "MYOLO_SPLIT_EVAL=0" "MYOLO_EVAL_BRANCH=0" "MYOLO_EVAL_ORDER=bc" "MYOLO_NO_STREAM=1" "MYOLO_NO_HALO=1" "A=1"; do
  for SZ in "1024 2048" "512 1024"; do
    R=$(env $E timeout 300 python bench.py --stage infer --infer-size $SZ --steps 300 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.0f FPS  stages %s' % (j['value'], {k: (round(v, 3) if isinstance(v, float) else v) for k, v in j.get('stage_ms', {}).items() if k != 'what'}))" 2>&1 | tail -1)
    echo "[$E] infer $SZ: $R" | tee -a gpurun_out/r5g_infer_sweep.txt
  done
done
for E in "A=0" "MYOLO_WGRAD_WG_TAIL=256" "MYOLO_WGRAD_WG_TAIL=192 MYOLO_WGRAD_TAIL_FRAC=0.3" "MYOLO_WGRAD_TAIL_FRAC=0.05" "MYOLO_WGRAD_WG_HINT=112" "A=1"; do
  R=$(env $E timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-infer --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.3f ms  %.0f img/s' % (j['ms_per_step'], j['value']))" 2>&1 | tail -1)
  echo "[$E] train: $R" | tee -a gpurun_out/r5g_train_sweep.txt
done
