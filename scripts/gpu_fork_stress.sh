# the un-joined eval forward under stress (scripts/ubench/fork_stress.py: every frame against the eager launch list) + the FPS of the fork mechanisms
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_queue_sem.py tests/test_gpu_model.py tests/test_gpu_postproc.py tests/test_gpu_dropin.py -q -m gpu -k "queue or sem or unjoined or eval or unchanged or detect" 2>&1 | tail -4
for F in sem sem event joined; do for T in f32 f16; do FORK=$F python scripts/ubench/fork_stress.py $T 2>&1 | grep -v "Fusing\|amdgpu.ids" | tail -2 | cut -c1-200; done; done 2>&1 | tee gpurun_out/fork_stress.txt
for F in event sem event sem; do
  for S in "1024 2048" "512 1024"; do
    echo "FORK=$F $S: $(timeout 300 python bench.py --eval-fork $F --stage infer --infer-size $S --steps 300 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'unchanged caller', round(d['unchanged_caller']['value'],1), {k: round(v, 3) for k, v in d['stage_ms'].items() if isinstance(v, float)})")"
  done
done 2>&1 | tee gpurun_out/infer_fork_ab.txt
