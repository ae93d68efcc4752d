cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dbg
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dbg/wg -o x -- python scripts/wgrad_bench.py > /dev/null 2>&1
python - <<PY
import csv,glob,statistics
f=glob.glob('gpurun_out/dbg/wg/**/x_kernel_stats.csv',recursive=True)+glob.glob('gpurun_out/dbg/wg/x_kernel_stats.csv')
for r in csv.DictReader(open(f[0])):
    if 'wgrad' in r['Name']: print(r['Name'][:60], r['Calls'], 'avg us', float(r['AverageNs'])/1e3, 'min', float(r['MinNs'])/1e3, 'max', float(r['MaxNs'])/1e3)
PY
timeout 900 python -m pytest tests -m gpu -q --timeout 200 2>&1 | tail -3 | cut -c1-300
echo "--- bench"; timeout 600 python bench.py --steps 20 --warmup 5 --no-infer --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
