cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_upce.py tests/test_gpu_dist.py -m gpu -q -x --timeout 900 2>&1 | tail -5 | cut -c1-300
for v in "MYOLO_STAGED_BWD=1" "MYOLO_STAGED_BWD=force" "MYOLO_NATIVE_EXEC=0"; do
echo -n "$v: "; env $v timeout 600 python bench.py --steps 40 --warmup 10 --no-infer --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | cut -c100-260
done
bash scripts/gpu_trace.sh r3e 2>&1 | grep -E "step:|queue|GPU busy" | cut -c1-200
