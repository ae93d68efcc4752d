# round-4 measurement run: smoke, the bench line, rocprofv3 stats + the two PMC passes of the same command, the per-layer conv table,
# inference profiles at both sizes, yolov5m + Lab, 2 ranks on one GPU through `bench.py --gpus 2` itself.  usage: bash scripts/gpu_r4_final.sh <tag>
TAG=${1:-r4}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "--- smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-400
echo "--- bench"; timeout 1200 python bench.py > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log | cut -c1-3000
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o train -- python bench.py --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/prof_$TAG -name 'train_kernel_stats.csv' | head -1) $TAG 7 gpurun_out/bench_$TAG.log
python scripts/conv_trace.py $(find gpurun_out/prof_$TAG -name 'train_kernel_trace.csv' | head -1) > gpurun_out/${TAG}_conv_layers.txt 2>&1; head -4 gpurun_out/${TAG}_conv_layers.txt
python scripts/trace_timeline.py $(find gpurun_out/prof_$TAG -name 'train_kernel_trace.csv' | head -1) > gpurun_out/${TAG}_timeline.txt 2>&1; head -4 gpurun_out/${TAG}_timeline.txt
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$C -o x -- python bench.py --steps 2 --warmup 1 --no-kernel-timing --no-infer --no-cpu-baseline > gpurun_out/pmc_${TAG}_$C.log 2>&1
  tail -1 gpurun_out/pmc_${TAG}_$C.log | cut -c1-120
done
python scripts/pmc_summary.py $TAG 3 2>&1 | tail -8
bash scripts/gpu_inf.sh $TAG 2>&1 | grep -E "^\{" | cut -c1-700
bash scripts/gpu_mlab.sh $TAG 2>&1 | grep -E "^\{" | cut -c1-400
cp profiles/${TAG}_* gpurun_out/ 2>/dev/null
echo "--- bench.py --gpus 2 (two ranks on this one GPU, gloo)"; MYOLO_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --batch 4 --no-cpu-baseline --no-infer --no-kernel-timing 2>&1 | tail -1 | cut -c1-700
echo "--- host time"; timeout 300 python scripts/host_time.py 2>&1 | tail -3 | cut -c1-300
