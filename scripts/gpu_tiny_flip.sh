# next step for csrc/tiny_conv.hip (second session of round 4): the WHOLE GPU suite with MYOLO_TINY_CONV=1 (serial: under pytest-xdist
# six processes on one GPU were ~5x slower than one), then the step off / on / off / on with the losses of each run, then kernel stats with it on.
# If the suite is green and the on / off loss gap is inside the off / off spread over more runs: flip the default in engine.py (TINY_CONV).
# usage: bash scripts/gpu_tiny_flip.sh [tag]
TAG=${1:-tinyflip}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "--- suite, MYOLO_TINY_CONV=1"; MYOLO_TINY_CONV=1 timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/suite_$TAG.log 2>&1; tail -6 gpurun_out/suite_$TAG.log | cut -c1-300
for E in "MYOLO_TINY_CONV=0" "MYOLO_TINY_CONV=1" "MYOLO_TINY_CONV=0" "MYOLO_TINY_CONV=1" "MYOLO_TINY_CONV=0" "MYOLO_TINY_CONV=1"; do
  R=$(env $E timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-infer --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.3f ms  %.0f img/s' % (j['ms_per_step'], j['value']), j['checks'])" 2>&1 | tail -1)
  echo "[$E]: $R" | tee -a gpurun_out/${TAG}_step.txt
done
MYOLO_TINY_CONV=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o train -- python bench.py --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1
python scripts/trace_timeline.py $(find gpurun_out/prof_$TAG -name 'train_kernel_trace.csv' | head -1) > gpurun_out/${TAG}_timeline.txt 2>&1; head -12 gpurun_out/${TAG}_timeline.txt
