cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -3 | cut -c1-300
echo "--- bench"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['achieved'], j.get('detect_fps',{}).get('value'))"
