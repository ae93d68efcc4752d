cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dbg
run() { # name, env...
  name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dbg/$name -o x -- python scripts/conv_bench3.py $MODE > /dev/null 2>&1
  python - <<PY
import csv,glob,statistics
f=glob.glob('gpurun_out/dbg/$name/**/x_kernel_trace.csv',recursive=True)+glob.glob('gpurun_out/dbg/$name/x_kernel_trace.csv')
rows=list(csv.DictReader(open(f[0])))
c=[r for r in rows if 'conv_stream' in r['Kernel_Name'] or 'conv_igemm' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in c]
a=d[5:15]; b=d[20:30]
print('$name $MODE', 'shape1 %.1f us  shape2 %.1f us' % (statistics.median(a), statistics.median(b)), c[10]['Kernel_Name'][:60], 'vgpr', c[10]['VGPR_Count'])
PY
}
for MODE in eval train; do
export MODE
run full_$MODE A=1
true
run v1_$MODE MYOLO_NO_STREAM=1
done
MYOLO_STREAM_MIN_TILES=1 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q --timeout 200 -k "f16" 2>&1 | tail -3 | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q --timeout 200 -k "f32" 2>&1 | tail -3 | cut -c1-300
echo "--- bench"; timeout 600 python bench.py --steps 20 --warmup 5 --no-infer --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
