cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_dropin.py tests/test_gpu_upce.py -x -q 2>&1 | tail -4
for v in "MYOLO_EVAL_BRANCH=0" "X=1" "MYOLO_EVAL_BRANCH=0" "X=1"; do
echo -n "infer $v: "; env $v timeout 300 python bench.py --stage infer --no-cpu-baseline --steps 100 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value'],1), round(j['stage_ms']['forward'],4), j['graph_replayed'])"
echo -n "infer1024 $v: "; env $v timeout 300 python bench.py --stage infer --infer-size 512 1024 --no-cpu-baseline --steps 100 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value'],1), round(j['stage_ms']['forward'],4), j['graph_replayed'])"
done
