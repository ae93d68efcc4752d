cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dbg
run() { name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dbg/$name -o x -- python scripts/conv_bench3.py $MODE > /dev/null 2>&1
  python - <<PY
import csv,glob,statistics
f=glob.glob('gpurun_out/dbg/$name/**/x_kernel_trace.csv',recursive=True)+glob.glob('gpurun_out/dbg/$name/x_kernel_trace.csv')
rows=list(csv.DictReader(open(f[0])))
c=[r for r in rows if 'conv_stream' in r['Kernel_Name'] or 'conv_igemm' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in c]
print('$name $MODE', ' | '.join('%.1f us (%s,%s)' % (statistics.median(d[i*15+5:i*15+15]), int(c[i*15+6]['Grid_Size_X'])//int(c[i*15+6]['Workgroup_Size_X']), c[i*15+6]['Grid_Size_Y']) for i in range(len(d)//15)))
PY
}
MODE=train
export MODE
run full A=1
run nostore MYOLO_STREAM_DBG=1
run noload MYOLO_STREAM_DBG=2
run neither MYOLO_STREAM_DBG=3
