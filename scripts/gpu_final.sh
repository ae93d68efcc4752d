# end-of-round sanity on the GPU box: parity tests, smoke(), the bench line, and the N>1 code path (2 ranks on ONE GPU over gloo)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -3 | cut -c1-300
echo "--- smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-300
echo "--- bench"; timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_final.log 2>&1; tail -1 gpurun_out/bench_final.log | cut -c1-1800
echo "--- 2 ranks on one GPU (gloo)"; MYOLO_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --batch 4 --no-cpu-baseline --no-infer --no-kernel-timing 2>&1 | tail -1 | cut -c1-500
