cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dbg
run() { name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dbg/$name -o x -- python scripts/wgrad_bench.py > /dev/null 2>&1
  python - <<PY
import csv,glob,statistics
f=glob.glob('gpurun_out/dbg/$name/**/x_kernel_trace.csv',recursive=True)+glob.glob('gpurun_out/dbg/$name/x_kernel_trace.csv')
rows=list(csv.DictReader(open(f[0])))
for kn in ('wgrad_kernel','wgrad_reduce'):
    c=[r for r in rows if kn in r['Kernel_Name']]
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in c]
    print('$name',kn, ' | '.join('%.1f us' % (statistics.median(d[i*9+3:i*9+9])) for i in range(3)))
PY
}
run wg_full A=1
timeout 900 python -m pytest tests -m gpu -q --timeout 200 2>&1 | tail -3 | cut -c1-300
echo "--- bench"; timeout 600 python bench.py --steps 20 --warmup 5 --no-infer --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
