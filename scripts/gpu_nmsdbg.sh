cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python - <<PY
import torch, time, sys
sys.path.insert(0,'.')
from multiyolov5_amd import synth, _lib as L
from multiyolov5_amd.utils.general import non_max_suppression
dev=torch.device('cuda:0')
for A,wh in ((32256,(1024,512)),(129024,(2048,1024))):
    pred=synth.nms_pred(1,A,10,seed=3,img_w=wh[0],img_h=wh[1]).to(dev,torch.float16)
    for dbg in (0,1,2,4,6,8,14):
        L.lib().myolo_set_option(b'nms_dbg',dbg)
        for _ in range(3): non_max_suppression(pred,0.25,0.45)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(30): o=non_max_suppression(pred,0.25,0.45)
        torch.cuda.synchronize(); print(A,'dbg',dbg,'%.1f us'%((time.perf_counter()-t0)/30*1e6), o[0].shape[0])
PY
