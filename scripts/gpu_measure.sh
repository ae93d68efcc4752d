TAG=${1:-r2b}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.log 2>&1
tail -1 gpurun_out/bench_$TAG.log | cut -c1-3000
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o train -- python bench.py --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1
tail -1 gpurun_out/prof_$TAG.log | cut -c1-200
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$C -o x -- python bench.py --steps 2 --warmup 1 --no-kernel-timing --no-infer --no-cpu-baseline > gpurun_out/pmc_${TAG}_$C.log 2>&1
  tail -1 gpurun_out/pmc_${TAG}_$C.log | cut -c1-120
done
ls gpurun_out/prof_$TAG gpurun_out/pmc_${TAG}_FETCH_SIZE | head
