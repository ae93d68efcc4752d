cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp multiyolov5_amd/lib/libmyolo.so /tmp/new.so
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -3
for rep in 1 2; do
for v in "igemm_old" "X=1"; do
echo -n "$v: "; env $( [ "$v" = igemm_old ] && echo MYOLO_LIB=$GRAFT_REPO_ROOT/multiyolov5_amd/lib/libmyolo_prev.so || echo X=1) timeout 300 python bench.py --steps 40 --warmup 10 --no-infer --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; print('step ms', json.loads(sys.stdin.read())['ms_per_step'])"
done; done
for v in "X=1" ; do
echo -n "infer $v: "; env $v timeout 300 python bench.py --stage infer --no-cpu-baseline --steps 100 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value'],1), round(j['stage_ms']['forward'],4))"
echo -n "infer1024 $v: "; env $v timeout 300 python bench.py --stage infer --infer-size 512 1024 --no-cpu-baseline --steps 100 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value'],1), round(j['stage_ms']['forward'],4))"
done
