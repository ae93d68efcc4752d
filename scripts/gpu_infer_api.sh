cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for F in sem event; do
  MYOLO_EVAL_FORK=$F timeout 300 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d gpurun_out/itr_$F -o tr -- python bench.py --stage infer --infer-size 1024 2048 --steps 40 --no-cpu-baseline > gpurun_out/api_$F.log 2>&1
  ls gpurun_out/itr_$F/* | head
  python scripts/trace_api_list.py $(find gpurun_out/itr_$F -name "*kernel_trace.csv" | head -1) $(find gpurun_out/itr_$F -name "*hip_api_trace.csv" | head -1) > gpurun_out/infer_api_listing_$F.txt 2>&1
  rm -rf gpurun_out/itr_$F
  wc -l gpurun_out/infer_api_listing_$F.txt
done
timeout 300 python -m pytest tests/test_gpu_postproc.py -q -m gpu -x 2>&1 | tail -30
