cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_postproc.py -m gpu -q --timeout 200 2>&1 | tail -2
python bench.py --stage infer 2>&1 | tail -1
