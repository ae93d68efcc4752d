# detect.py frame loop: kernel traces at both sizes (per-frame idle time of the GPU, scripts/trace_infer_timeline.py), host time per phase,
# then the FPS under each environment given as an argument (e.g. "MYOLO_EVAL_TAIL=e" "MYOLO_EVAL_ORDER=bc" "ROC_ACTIVE_WAIT_TIMEOUT=2000").
# usage: bash scripts/gpu_infer_timeline.sh ["ENV=a ENV2=b" ...]      (-> profiles/r3k_infer_timeline.md)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for S in "1024 2048" "512 1024"; do
  T=$(echo $S | tr ' ' 'x')
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/itrace_$T -o tr -- python bench.py --stage infer --infer-size $S --steps 60 --no-cpu-baseline > gpurun_out/itrace_$T.log 2>&1
  python scripts/trace_infer_timeline.py $(find gpurun_out/itrace_$T -name "*kernel_trace.csv" | head -1) 2>&1 | tail -20
done
echo "--- host time (1024x512)"; timeout 200 python scripts/host_time_infer.py 2>&1 | head -30 | cut -c1-160
for V in "" "$@"; do
  for S in "1024 2048" "512 1024"; do
    echo "== [$V] $S: $(env $V timeout 120 python bench.py --stage infer --infer-size $S --steps 300 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'FPS', {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['stage_ms'].items() if k!='what'})")"
  done
done
