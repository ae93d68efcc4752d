# HIP / ROCr runtime switches against the two-queue launch pattern (scripts/ubench/two_queue_gap.py): which one owns the ~90 us?
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for E in "X=1" "ROC_CPU_WAIT_FOR_SIGNAL=0" "ROC_CPU_WAIT_FOR_SIGNAL=1" "ROC_ACTIVE_WAIT_TIMEOUT=1000" "ROC_ACTIVE_WAIT_TIMEOUT=0" "ROC_SYSTEM_SCOPE_SIGNAL=0" "GPU_STREAMOPS_CP_WAIT=1" "GPU_STREAMOPS_CP_WAIT=0" \
         "DEBUG_HIP_FORCE_ASYNC_QUEUE=1" "DEBUG_HIP_DYNAMIC_QUEUES=1" "DEBUG_HIP_DYNAMIC_QUEUES=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=8" \
         "HSA_ENABLE_INTERRUPT=0" "DEBUG_HIP_BLOCK_SYNC=0" "DEBUG_HIP_BLOCK_SYNC=1" "AMD_DIRECT_DISPATCH=0" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=4"; do
  echo "=== $E"
  env $E timeout 120 python scripts/ubench/two_queue_gap.py 2>&1 | grep -v amdgpu.ids | cut -c1-120
done > gpurun_out/env_sweep.txt 2>&1
grep -c us gpurun_out/env_sweep.txt
