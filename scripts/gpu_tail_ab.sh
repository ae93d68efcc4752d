# in-step A/B of the weight-gradient workgroup target for the LAST layers of the backward (engine.WGRAD_WG_TAIL / WGRAD_TAIL_FRAC), same box
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2; do for cfg in "0 0.15" "192 0.05" "256 0.05" "384 0.05" "256 0.1" "512 0.05"; do
  set -- $cfg
  timeout 300 python -c "
import sys, runpy
sys.path.insert(0, '.')
import multiyolov5_amd.engine as E
E.WGRAD_WG_TAIL, E.WGRAD_TAIL_FRAC = $1, $2
sys.argv = ['bench.py', '--steps', '40', '--warmup', '10', '--no-cpu-baseline', '--no-infer', '--no-kernel-timing', '--no-stock-baseline']
runpy.run_path('bench.py', run_name='__main__')
" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tail_wg=$1 frac=$2', round(d['ms_per_step'],3), round(d['value'],1))"
done; done 2>&1 | tee gpurun_out/tail_ab.txt
