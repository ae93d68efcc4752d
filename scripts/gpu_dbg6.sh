cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 300 -x -k "conv3x3_small" 2>&1 | tail -5 | cut -c1-400
for v in 1 0; do echo -n "small=$v: "; if [ $v = 0 ]; then export MYOLO_WGRAD_NO_SMALL=1; fi; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"; done
