# BASELINE configs[3] per-GPU share: yolov5m + Lab head, bs 8: bench line with the conv roofline + rocprofv3 kernel table -> profiles/<tag>_mlab_*
TAG=${1:-r3}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --cfg yolov5m_city_seg_lab.yaml --batch 8 --steps 30 --warmup 8 --no-infer --no-cpu-baseline > gpurun_out/bench_${TAG}_mlab.log 2>&1
tail -1 gpurun_out/bench_${TAG}_mlab.log | cut -c1-1500
CMD="rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --cfg yolov5m_city_seg_lab.yaml --batch 8 --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_mlab -o train -- python bench.py --cfg yolov5m_city_seg_lab.yaml --batch 8 --steps 5 --warmup 2 --no-kernel-timing --no-infer --no-cpu-baseline > gpurun_out/prof_${TAG}_mlab.log 2>&1
F=$(find gpurun_out/prof_${TAG}_mlab -name 'train_kernel_stats.csv' | head -1)
python scripts/prof_summary.py $F ${TAG}_mlab 7 gpurun_out/bench_${TAG}_mlab.log step "$CMD"
cp profiles/${TAG}_mlab_* gpurun_out/
python scripts/trace_timeline.py $(find gpurun_out/prof_${TAG}_mlab -name "*kernel_trace.csv" | head -1) | head -34 | cut -c1-160
