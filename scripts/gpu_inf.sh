# detect.py path (BASELINE configs[4] + the 1024x512 point): bench line (hipGraph replay) + rocprofv3 kernel trace of the same command
# usage: bash scripts/gpu_inf.sh <tag>      -> profiles/<tag>_infer{2048,1024}_{summary.md,kernel_stats.csv}
TAG=${1:-r3a}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for SZ in "1024 2048" "512 1024"; do
  set -- $SZ; H=$1; W=$2
  timeout 600 python bench.py --stage infer --infer-size $H $W --steps 60 > gpurun_out/infer_${TAG}_$W.log 2>&1
  tail -1 gpurun_out/infer_${TAG}_$W.log | cut -c1-1500
  CMD="rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --stage infer --infer-size $H $W --steps 60 --no-cpu-baseline"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_infer$W -o inf -- python bench.py --stage infer --infer-size $H $W --steps 60 --no-cpu-baseline > gpurun_out/prof_${TAG}_infer$W.log 2>&1
  F=$(find gpurun_out/prof_${TAG}_infer$W -name 'inf_kernel_stats.csv' | head -1)
  python scripts/prof_summary.py $F ${TAG}_infer$W 78 gpurun_out/infer_${TAG}_$W.log frame "$CMD"
  python - <<PY
import csv
rows=list(csv.DictReader(open('$F')))
frames=78
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('$W: total kernel ms/frame', tot/frames/1e6)
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls/frame {int(r['Calls'])/frames:6.1f} avg_us {float(r['AverageNs'])/1e3:8.1f} ms/frame {float(r['TotalDurationNs'])/frames/1e6:6.3f}")
PY
done
cp profiles/${TAG}_infer* gpurun_out/ 2>/dev/null
