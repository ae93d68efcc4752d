cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/r2j_tests.log 2>&1
tail -6 gpurun_out/r2j_tests.log | cut -c1-300
echo "--- bench eager full"; MYOLO_GRAPH_TRAIN=0 timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r2j_bench_eager.log 2>&1; tail -1 gpurun_out/r2j_bench_eager.log | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['detect_fps'], j['train_py_step']['pairs_per_s'], j['cpu_baseline'])"
echo "--- bench graph"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer --no-kernel-timing 2>/dev/null | tail -1 | cut -c1-250
