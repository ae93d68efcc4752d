cd $GRAFT_REPO_ROOT
for mt in 2048 1024 512 256; do echo "min_tiles=$mt"; MYOLO_STREAM_MIN_TILES=$mt timeout 300 python bench.py --steps 20 --warmup 5 --no-infer --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done
