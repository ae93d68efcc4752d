# stock PyTorch-ROCm (ATen + MIOpen) run of the same graph on the GPU box; partial results survive in gpurun_out/stock.log
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CL=${1:-}
export MIOPEN_FIND_MODE=FAST MIOPEN_USER_DB_PATH=/tmp/miopen MIOPEN_LOG_LEVEL=2
(timeout 170 python -m oracle.rocm_stock_bench --steps 8 --warmup 2 $CL --no-infer; \
 timeout 60 python -m oracle.rocm_stock_bench --steps 8 --warmup 2 $CL --no-train) > gpurun_out/stock.log 2>&1
grep -v "^MIOpen\|Warning" gpurun_out/stock.log | tail -8 | cut -c1-600
