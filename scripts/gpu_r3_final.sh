# round-3 measurement run: full parity suite, smoke, the bench line, rocprofv3 stats + the two PMC passes of the same command, inference
# profiles at both sizes, 2 ranks on one GPU.  usage: bash scripts/gpu_r3_final.sh <tag>
TAG=${1:-r3h}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -3 | cut -c1-300
echo "--- smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-300
echo "--- bench"; timeout 900 python bench.py > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log | cut -c1-2500
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o train -- python bench.py --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/prof_$TAG -name 'train_kernel_stats.csv' | head -1) $TAG 7 gpurun_out/bench_$TAG.log
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$C -o x -- python bench.py --steps 2 --warmup 1 --no-kernel-timing --no-infer --no-cpu-baseline > gpurun_out/pmc_${TAG}_$C.log 2>&1
  tail -1 gpurun_out/pmc_${TAG}_$C.log | cut -c1-120
done
python scripts/pmc_summary.py $TAG 3 2>&1 | tail -3
bash scripts/gpu_inf.sh $TAG 2>&1 | grep -E "^\{" | cut -c1-700
cp profiles/${TAG}_* gpurun_out/ 2>/dev/null
echo "--- 2 ranks on one GPU (gloo)"; MYOLO_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --batch 4 --no-cpu-baseline --no-infer --no-kernel-timing 2>&1 | tail -1 | cut -c1-700
echo "--- host time"; timeout 300 python scripts/host_time.py 2>&1 | tail -3 | cut -c1-300
