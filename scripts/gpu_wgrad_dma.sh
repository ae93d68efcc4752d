# LDS-DMA loaders of wgrad_tile (round 4): per-layer A/B + numerics of every variant, the real-shape fp16 layer tests, the step under both loaders.
# usage: bash scripts/gpu_wgrad_dma.sh [tag]
TAG=${1:-wdma}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "--- ubench (all shapes: dma | reg)"; timeout 300 python scripts/wgrad_ubench.py > gpurun_out/${TAG}_ubench.txt 2>&1; tail -22 gpurun_out/${TAG}_ubench.txt | cut -c1-260
echo "--- ubench sweep (quick shapes)"; timeout 300 python scripts/wgrad_ubench.py quick sweep > gpurun_out/${TAG}_ubench_sweep.txt 2>&1; tail -8 gpurun_out/${TAG}_ubench_sweep.txt | cut -c1-300
echo "--- real-shape layer tests (train)"; timeout 600 python -m pytest tests/test_gpu_configs.py -q -x -k "real_shape and train" 2>&1 | tail -4 | cut -c1-300
for E in "MYOLO_WGRAD_TILE_DMA=0" "MYOLO_WGRAD_TILE_DMA=1" "MYOLO_WGRAD_TILE_DMA=1 MYOLO_WGRAD_TILE_MIN_TILES=3" "MYOLO_WGRAD_TILE_DMA=1 MYOLO_WGRAD_TILE_MIN_TILES=4 MYOLO_WGRAD_TILE_WG=192"; do
  R=$(env $E timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-infer --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.3f ms  %.0f img/s' % (j['ms_per_step'], j['value']))")
  echo "[$E]: $R" | tee -a gpurun_out/${TAG}_step.txt
done
