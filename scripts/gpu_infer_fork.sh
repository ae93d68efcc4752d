# detect.py frame: semaphore fork against the event fork, same box, alternating; chronological listing of one frame each
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_queue_sem.py tests/test_gpu_model.py -q -m gpu -k "queue or sem or unjoined or eval" 2>&1 | tail -5
for F in event sem event sem; do
  for S in "1024 2048" "512 1024"; do
    echo "MYOLO_EVAL_FORK=$F $S: $(MYOLO_EVAL_FORK=$F timeout 300 python bench.py --stage infer --infer-size $S --steps 300 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'unchanged caller', round(d['unchanged_caller']['value'],1), d['stage_ms'])")"
  done
done 2>&1 | tee gpurun_out/infer_fork_ab.txt
for F in event sem; do
  MYOLO_EVAL_FORK=$F timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/itr_$F -o tr -- python bench.py --stage infer --infer-size 1024 2048 --steps 40 --no-cpu-baseline > /dev/null 2>&1
  python scripts/trace_list.py $(find gpurun_out/itr_$F -name "*kernel_trace.csv" | head -1) seg_argmax > gpurun_out/infer_fork_listing_$F.txt 2>&1
  rm -rf gpurun_out/itr_$F
  grep "gap  *[1-9][0-9]\.[0-9]\|gap  *[1-9][0-9][0-9]\.[0-9]\|queue_\|seg_argmax" gpurun_out/infer_fork_listing_$F.txt | cut -c1-110
done
