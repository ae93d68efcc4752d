# A/B of an environment switch in ONE session: bash scripts/gpu_ab.sh VAR valueA valueB
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in $2 $3; do
echo -n "$1=$v: "; env $1=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
done; done
