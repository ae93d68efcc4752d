cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
MYOLO_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_inf -o inf -- python bench.py --stage infer > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_inf/inf_kernel_stats.csv')))
frames=35
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms/frame', tot/frames/1e6)
for r in rows[:16]:
    print(f"{r['Name'][:60]:60s} calls/frame {int(r['Calls'])/frames:6.1f} avg_us {float(r['AverageNs'])/1e3:8.1f} ms/frame {float(r['TotalDurationNs'])/frames/1e6:6.3f}")
PY
