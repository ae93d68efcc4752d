# detect.py frame loop: kernel traces at both sizes (per-frame idle time of the GPU) + host time per phase.  usage: bash scripts/gpu_r3k.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for S in "1024 2048" "512 1024"; do
  T=$(echo $S | tr ' ' 'x')
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/itrace_$T -o tr -- python bench.py --stage infer --infer-size $S --steps 60 --no-cpu-baseline > gpurun_out/itrace_$T.log 2>&1
  tail -1 gpurun_out/itrace_$T.log | cut -c1-400
  python scripts/trace_infer_timeline.py $(find gpurun_out/itrace_$T -name "*kernel_trace.csv" | head -1) 2>&1 | tail -20
done
echo "--- host time (1024x512)"; timeout 200 python scripts/host_time_infer.py 2>&1 | head -30 | cut -c1-160
