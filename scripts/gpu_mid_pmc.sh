# usage: bash scripts/gpu_mid_pmc.sh "cin,cout,k,s,d,H,W" [var]  -- SQ counters of conv_mid on one layer shape (rocprofv3 --pmc, kernel trace only)
SHAPE=${1:-128,128,3,1,1,32,64}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD"; do
  T=$(echo $SET | cut -c1-12 | tr ' ' '_')
  PROBE=one PROBE_SHAPE=$SHAPE MYOLO_CONV_MID=2 timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d gpurun_out/midpmc_$T -o x -- python scripts/conv_train_ubench.py > gpurun_out/midpmc_$T.log 2>&1
  tail -2 gpurun_out/midpmc_$T.log | cut -c1-200
  F=$(find gpurun_out/midpmc_$T -name "*counter_collection.csv" | head -1)
  python scripts/sq_pmc_summary.py $F conv_mid
done
