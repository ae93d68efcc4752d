cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_only.log 2>&1; tail -1 gpurun_out/bench_only.log | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(j['value'], j['ms_per_step'], j.get('detect_fps'), j.get('train_py_step'))"
