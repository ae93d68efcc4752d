cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for WG in 128 192; do
echo "=== new loader WG=$WG"; MYOLO_WGRAD_TILE_WG=$WG timeout 600 python scripts/wgrad_ubench.py quick 2>&1 | tail -5 | cut -c1-120
echo "=== old loader WG=$WG"; MYOLO_LIB=$GRAFT_REPO_ROOT/multiyolov5_amd/lib/libmyolo_ab.so MYOLO_WGRAD_TILE_WG=$WG timeout 600 python scripts/wgrad_ubench.py quick 2>&1 | tail -5 | cut -c1-120
done
MYOLO_LIB=$GRAFT_REPO_ROOT/multiyolov5_amd/lib/libmyolo_ab.so bash scripts/gpu_sweep.sh "MYOLO_OLDLOADER=1"
