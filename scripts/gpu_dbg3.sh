cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python scripts/segce_bench.py 2>&1 | tail -3
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x 2>&1 | tail -12 | cut -c1-300
echo "--- bench"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['achieved'], j.get('detect_fps',{}).get('value'))"
