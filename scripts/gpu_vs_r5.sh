# same-box A/B of this tree against the round-5 tree checked out (and built) under _r5/: training step, alternating
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
F="--steps 40 --warmup 10 --no-cpu-baseline --no-infer --no-kernel-timing --no-stock-baseline"
for r in 1 2 3; do
  (cd _r5 && timeout 300 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r5   ', round(d['ms_per_step'],3), round(d['value'],1))")
  timeout 300 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r6   ', round(d['ms_per_step'],3), round(d['value'],1))"
  MYOLO_BN_BWD_FUSED=0 timeout 300 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r6 bnfused=0', round(d['ms_per_step'],3), round(d['value'],1))"
done 2>&1 | tee gpurun_out/vs_r5.txt
