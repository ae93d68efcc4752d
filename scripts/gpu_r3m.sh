# detect.py frame loop under host-wait settings of the HIP runtime (the per-frame NMS sync).  usage: bash scripts/gpu_r3m.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for V in "" "ROC_ACTIVE_WAIT_TIMEOUT=2000" "HSA_ENABLE_INTERRUPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=2000 HSA_ENABLE_INTERRUPT=0" "" "ROC_ACTIVE_WAIT_TIMEOUT=200"; do
  for S in "1024 2048" "512 1024"; do
    echo "== [$V] $S: $(env $V timeout 120 python bench.py --stage infer --infer-size $S --steps 400 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'FPS', {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['stage_ms'].items() if k!='what'})")"
  done
done
