cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py -m gpu -q -x --timeout 600 -k "wgrad or block or layer or config" 2>&1 | tail -3 | cut -c1-200
bash scripts/gpu_ab_lib.sh multiyolov5_amd/lib/ab/libmyolo_row1pipe.so 2>&1 | grep -v Traceback | cut -c1-125
