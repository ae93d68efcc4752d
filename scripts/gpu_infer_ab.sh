# usage: bash scripts/gpu_infer_ab.sh "ENV=a" "ENV=b" ...  -- detect.py path FPS (2048x1024 and 1024x512) under each environment, same box
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for E in "$@"; do
  for SZ in "1024 2048" "512 1024"; do
    R=$(env $E timeout 300 python bench.py --stage infer --infer-size $SZ --steps 200 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.0f FPS  stages %s' % (j['value'], j.get('stage_ms')))")
    echo "[$E] $SZ: $R" | tee -a gpurun_out/infer_ab.txt
  done
done
