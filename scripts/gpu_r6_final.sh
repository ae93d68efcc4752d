# end-of-round validation, round 6: whole GPU suite (fresh parity log), smoke, the bench line, rocprofv3 kernel stats + per-layer conv table + timeline +
# chronological listing of the training step, two PMC passes over 5 steady-state steps (plan build + warm-up dropped), the same for the yolov5m + Lab
# share, kernel stats + timeline of the detect.py frame.  usage: bash scripts/gpu_r6_final.sh <tag> [nosuite]
TAG=${1:-r6}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$2" != nosuite ]; then
  rm -f gpurun_out/parity_log.jsonl gpurun_out/bench_plan_variants.txt
  echo "--- suite"; (time timeout 1800 python -m pytest tests -m gpu -q --timeout 900) > gpurun_out/suite_$TAG.log 2>&1; tail -6 gpurun_out/suite_$TAG.log | cut -c1-300
  python scripts/parity_summary.py > gpurun_out/${TAG}_parity_summary.md 2>&1; head -3 gpurun_out/${TAG}_parity_summary.md | cut -c1-200
fi
echo "--- smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-400
echo "--- bench"; timeout 900 python bench.py > gpurun_out/bench_$TAG.log 2> gpurun_out/bench_$TAG.err; tail -1 gpurun_out/bench_$TAG.log | cut -c1-900
python -c "import json,sys; l=[x for x in open(sys.argv[1]).read().splitlines() if x.startswith(chr(123))][-1]; json.dump(json.loads(l), open(sys.argv[2], 'w'), indent=1)" gpurun_out/bench_$TAG.log gpurun_out/${TAG}_bench.json
echo "--- prof"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o train -- python bench.py --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1
TR=$(find gpurun_out/prof_$TAG -name 'train_kernel_trace.csv' | head -1)
python scripts/prof_summary.py $(find gpurun_out/prof_$TAG -name 'train_kernel_stats.csv' | head -1) $TAG 7 gpurun_out/bench_$TAG.log
python scripts/conv_trace.py $TR > gpurun_out/${TAG}_conv_layers.txt 2>&1; head -4 gpurun_out/${TAG}_conv_layers.txt
python scripts/trace_timeline.py $TR > gpurun_out/${TAG}_timeline.txt 2>&1; head -4 gpurun_out/${TAG}_timeline.txt
python scripts/trace_list.py $TR > gpurun_out/${TAG}_step_listing.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$C -o x -- python bench.py --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline --no-stock-baseline > gpurun_out/pmc_${TAG}_$C.log 2>&1
  tail -1 gpurun_out/pmc_${TAG}_$C.log | cut -c1-120
done
python scripts/pmc_summary.py $TAG 2 2>&1 | tail -10
echo "--- yolov5m + Lab (BASELINE configs[3], per-GPU share)"
timeout 600 python bench.py --cfg yolov5m_city_seg_lab.yaml --batch 8 --steps 30 --warmup 8 --no-infer --no-cpu-baseline --no-stock-baseline > gpurun_out/bench_${TAG}_mlab.log 2>/dev/null
tail -1 gpurun_out/bench_${TAG}_mlab.log | cut -c1-400
CMD="rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --cfg yolov5m_city_seg_lab.yaml --batch 8 --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_mlab -o train -- python bench.py --cfg yolov5m_city_seg_lab.yaml --batch 8 --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline > gpurun_out/prof_${TAG}_mlab.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/prof_${TAG}_mlab -name 'train_kernel_stats.csv' | head -1) ${TAG}_mlab 7 gpurun_out/bench_${TAG}_mlab.log step "$CMD"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_mlab_$C -o x -- python bench.py --cfg yolov5m_city_seg_lab.yaml --batch 8 --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline --no-stock-baseline > gpurun_out/pmc_${TAG}_mlab_$C.log 2>&1
done
python scripts/pmc_summary.py ${TAG}_mlab 2 2>&1 | tail -8
echo "--- frame"
for S in "1024 2048" "512 1024"; do
  T=$(echo $S | tr ' ' 'x')
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_itrace_$T -o tr -- python bench.py --stage infer --infer-size $S --steps 60 --no-cpu-baseline > gpurun_out/${TAG}_itrace_$T.log 2>&1
  python scripts/trace_infer_timeline.py $(find gpurun_out/${TAG}_itrace_$T -name "*kernel_trace.csv" | head -1) > gpurun_out/${TAG}_infer_timeline_$T.txt 2>&1
  python scripts/trace_list.py $(find gpurun_out/${TAG}_itrace_$T -name "*kernel_trace.csv" | head -1) seg_argmax > gpurun_out/${TAG}_infer_listing_$T.txt 2>&1
  head -4 gpurun_out/${TAG}_infer_timeline_$T.txt | cut -c1-200
  cp $(find gpurun_out/${TAG}_itrace_$T -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_infer${T}_kernel_stats.csv 2>/dev/null
  rm -rf gpurun_out/${TAG}_itrace_$T
done
echo "--- fork stress"
for F in sem event joined; do for T in f32 f16; do FORK=$F python scripts/ubench/fork_stress.py $T 2>&1 | grep -v "Fusing\|amdgpu.ids" | tail -1 | cut -c1-200; done; done 2>&1 | tee gpurun_out/${TAG}_fork_stress.txt
cp -n profiles/${TAG}_* gpurun_out/ 2>/dev/null     # (what prof_summary / pmc_summary wrote into profiles/ on the box travels back through gpurun_out; never over a file this run produced)
rm -rf gpurun_out/prof_$TAG gpurun_out/prof_${TAG}_mlab gpurun_out/pmc_${TAG}_* 2>/dev/null
du -sh gpurun_out | tail -1
