# usage: bash scripts/gpu_mid.sh [probe ...]  -- conv_mid parity tests + per-layer micro-benchmarks (PROBE names of scripts/conv_train_ubench.py)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_conv_mid.py -x -q 2>&1 | tail -15 | tee gpurun_out/mid_tests.txt
for P in "${@:-mid}"; do
  PROBE=$P timeout 600 python scripts/conv_train_ubench.py 2>&1 | grep -v amdgpu.ids | tail -40 | tee gpurun_out/mid_ubench_$P.txt
done
