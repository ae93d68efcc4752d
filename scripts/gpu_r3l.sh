# detect.py frame loop under the issue variants of the split eval forward (runtime.EVAL_TAIL / EVAL_HEAD / EVAL_ORDER).  usage: bash scripts/gpu_r3l.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for V in "" "MYOLO_EVAL_TAIL=e" "MYOLO_EVAL_HEAD=e" "MYOLO_EVAL_TAIL=e MYOLO_EVAL_HEAD=e" "MYOLO_EVAL_ORDER=bc" "MYOLO_EVAL_ORDER=bc MYOLO_EVAL_TAIL=e" "MYOLO_SPLIT_EVAL=0"; do
  for S in "1024 2048" "512 1024"; do
    echo "== [$V] $S: $(env $V timeout 120 python bench.py --stage infer --infer-size $S --steps 300 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'FPS', {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['stage_ms'].items() if k!='what'})")"
  done
done
V="MYOLO_EVAL_TAIL=e"
env $V timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/itrace_tail_e -o tr -- python bench.py --stage infer --infer-size 1024 2048 --steps 60 --no-cpu-baseline > gpurun_out/itrace_tail_e.log 2>&1
python scripts/trace_infer_timeline.py $(find gpurun_out/itrace_tail_e -name "*kernel_trace.csv" | head -1) 2>&1 | tail -8
