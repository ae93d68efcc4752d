cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in "X=1" "MYOLO_TRAIN_BRANCH=1" "X=1" "MYOLO_TRAIN_BRANCH=1"; do
echo -n "$v: "; env $v timeout 300 python bench.py --steps 40 --warmup 10 --no-infer --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('step ms', j['ms_per_step'], j['checks'])"
done
