cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_postproc.py -m gpu -q -x 2>&1 | tail -5
timeout 300 python bench.py --stage infer --infer-size 1024 2048 --steps 60 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-600
timeout 300 python bench.py --stage infer --infer-size 512 1024 --steps 60 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-600
timeout 300 python scripts/nms_bench.py 2>&1 | tail -12
