cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for sl in 0 64 128; do echo "=== slice $sl"; MYOLO_BN_SLICE=$sl timeout 300 python scripts/bn_ubench.py 2>&1 | tail -11; done
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -3
for v in "MYOLO_BN_SLICE=0" "MYOLO_BN_SLICE=64" "MYOLO_BN_SLICE=128" "MYOLO_BN_SLICE=0" "MYOLO_BN_SLICE=64" "MYOLO_BN_SLICE=128"; do
echo -n "$v: "; env $v timeout 300 python bench.py --steps 40 --warmup 10 --no-infer --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; print('step ms', json.loads(sys.stdin.read())['ms_per_step'])"
done
