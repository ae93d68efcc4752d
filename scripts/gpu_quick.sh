cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "wgrad or block" 2>&1 | tail -2
echo "=== new"; timeout 300 python scripts/wgrad_ubench.py quick 2>&1 | tail -6 | cut -c1-110
for v in "MYOLO_LIB=$GRAFT_REPO_ROOT/multiyolov5_amd/lib/libmyolo_prev.so" "X=1" "MYOLO_LIB=$GRAFT_REPO_ROOT/multiyolov5_amd/lib/libmyolo_prev.so" "X=1"; do
echo -n "$v: "; env $v timeout 300 python bench.py --steps 40 --warmup 10 --no-infer --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('step ms', round(j['ms_per_step'],3))"
done
