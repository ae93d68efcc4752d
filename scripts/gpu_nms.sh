cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_postproc.py -m gpu -q -x 2>&1 | tail -5
python - <<PY
import torch, time, sys
sys.path.insert(0,'.')
from multiyolov5_amd import synth, _lib as L
from multiyolov5_amd.utils.general import non_max_suppression
dev=torch.device('cuda:0')
for A,wh in ((32256,(1024,512)),(129024,(2048,1024))):
    pred=synth.nms_pred(1,A,10,seed=3,img_w=wh[0],img_h=wh[1]).to(dev,torch.float16)
    ref=None
    for dbg in (16,0):
        L.lib().myolo_set_option(b'nms_dbg',dbg)
        for _ in range(3): o=non_max_suppression(pred,0.25,0.45)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(30): o=non_max_suppression(pred,0.25,0.45)
        torch.cuda.synchronize(); print(A,'dbg',dbg,'%.1f us'%((time.perf_counter()-t0)/30*1e6), o[0].shape[0], 'same as lazy' if ref is not None and torch.equal(ref,o[0]) else '')
        ref=o[0].clone()
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nms -o nms -- python scripts/nms_bench.py detect > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_nms/**/nms_kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5} avg_us {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
timeout 300 python bench.py --stage infer --infer-size 1024 2048 --steps 60 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-500
timeout 300 python bench.py --stage infer --infer-size 512 1024 --steps 60 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-500
