# HBM traffic of the conv kernels: two separate PMC passes (TCC slot budget), kernel-trace only (MI355X_MICROARCH.md)
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$C -o x -- python bench.py --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline --no-stock-baseline > gpurun_out/pmc_${TAG}_$C.log 2>&1
  tail -1 gpurun_out/pmc_${TAG}_$C.log | cut -c1-200
  find gpurun_out/pmc_${TAG}_$C -name "*.csv" | head
done
