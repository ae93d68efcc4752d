cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "X=0" "MYOLO_WGRAD_TILE_LDS_KB=96" "MYOLO_WGRAD_TILE_LDS_KB=72" "MYOLO_WGRAD_TILE_LDS_KB=48" "MYOLO_WGRAD_TILE_WG=64" "MYOLO_WGRAD_TILE_WG=256" "MYOLO_WGRAD_TILE_LDS_KB=72 MYOLO_WGRAD_TILE_WG=256"; do
echo -n "$v: "; env $v timeout 600 python bench.py --steps 40 --warmup 10 --no-infer --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done; done
