cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "=== base"; MYOLO_LIB=$GRAFT_REPO_ROOT/multiyolov5_amd/lib/libmyolo_base.so timeout 300 python scripts/bn_ubench.py 2>&1 | tail -11
echo "=== new"; timeout 300 python scripts/bn_ubench.py 2>&1 | tail -11
echo "=== new caps 2048"; MYOLO_BN_CAP_FWD=2048 MYOLO_BN_CAP_APP=2048 MYOLO_BN_CAP_RED=1024 timeout 300 python scripts/bn_ubench.py 2>&1 | tail -11
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -3
for v in "MYOLO_LIB=$GRAFT_REPO_ROOT/multiyolov5_amd/lib/libmyolo_base.so" "X=1" "MYOLO_BN_CAP_FWD=2048 MYOLO_BN_CAP_APP=2048" "MYOLO_LIB=$GRAFT_REPO_ROOT/multiyolov5_amd/lib/libmyolo_base.so" "X=1"; do
echo -n "$v: "; env $v timeout 300 python bench.py --steps 40 --warmup 10 --no-infer --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; print('step ms', json.loads(sys.stdin.read())['ms_per_step'])"
done
for v in "MYOLO_LIB=$GRAFT_REPO_ROOT/multiyolov5_amd/lib/libmyolo_base.so" "X=1"; do
echo -n "infer $v: "; env $v timeout 300 python bench.py --stage infer --no-cpu-baseline --steps 100 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value'],1), round(j['stage_ms']['forward'],4))"
done
