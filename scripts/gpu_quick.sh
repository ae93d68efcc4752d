cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q 2>&1 | tail -5
for v in "MYOLO_NO_BN96=1" "X=1" "MYOLO_NO_BN96=1" "X=1"; do
echo -n "mlab $v: "; env $v timeout 300 python bench.py --cfg yolov5m_city_seg_lab.yaml --batch 8 --steps 30 --warmup 8 --no-infer --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('step ms', j['ms_per_step'], j['value'])"
done
