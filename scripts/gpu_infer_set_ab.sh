# detect.py frame: A/B of myolo_set_option settings on one box, alternating.  usage: bash scripts/gpu_infer_set_ab.sh "name=value X=1" [rounds]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${2:-3}
for i in $(seq $R); do
  for E in $1; do
    for S in "1024 2048" "512 1024"; do
      echo "$E $S: $(MYOLO_SET=$E python bench.py --stage infer --infer-size $S --steps 300 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['stage_ms']['forward_main_chain'],3))")"
    done
  done
done
