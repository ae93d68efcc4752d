# A/B of differently built libraries: bash scripts/gpu_ab_lib.sh <ubench args> -- lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for lib in default "$@"; do
  echo "=== $lib"
  if [ "$lib" = default ]; then unset MYOLO_LIB; else export MYOLO_LIB=$GRAFT_REPO_ROOT/$lib; fi
  timeout 600 python scripts/wgrad_ubench.py 2>&1 | grep -E "k3|k1" | cut -c1-120
  timeout 600 python bench.py --steps 40 --warmup 10 --no-infer --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; print('step ms', json.loads(sys.stdin.read())['ms_per_step'])"
done
