cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nms -o nms -- python scripts/nms_bench.py detect > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_nms/**/nms_kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5} avg_us {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
