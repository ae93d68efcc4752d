# A/B of environment settings in ONE session: bash scripts/gpu_ab2.sh "A=1 B=2" "A=3" ...   (each arg = one env assignment list)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "$@"; do
echo -n "$v: "; env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
done; done
