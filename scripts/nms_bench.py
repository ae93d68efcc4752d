"""non_max_suppression timing in the two regimes the reference uses: detect.py (conf 0.25, iou 0.45, best class) on a 2048x1024 frame's
129 024 candidates and test.py (conf 0.001, iou 0.6, multi_label: up to 30 000 sorted candidates) on 32 256.  usage: python scripts/nms_bench.py [substring of a regime name]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiyolov5_amd import synth
from multiyolov5_amd.utils.general import non_max_suppression

dev = torch.device('cuda:0')
for name, A, wh, dt, kw in (('detect.py 2048x1024 fp16', 129024, (2048, 1024), torch.float16, dict(conf_thres=0.25, iou_thres=0.45)),
                            ('detect.py 1024x512 fp16', 32256, (1024, 512), torch.float16, dict(conf_thres=0.25, iou_thres=0.45)),
                            ('test.py 1024x512 fp32 multi_label', 32256, (1024, 512), torch.float32, dict(conf_thres=0.001, iou_thres=0.6, multi_label=True)),
                            ('test.py batch 8', 32256, (1024, 512), torch.float32, dict(conf_thres=0.001, iou_thres=0.6, multi_label=True))):
    if len(sys.argv) > 1 and sys.argv[1] not in name:
        continue
    B = 8 if 'batch 8' in name else 1
    pred = synth.nms_pred(B, A, 10, seed=3, img_w=wh[0], img_h=wh[1]).to(dev, dt)
    for _ in range(3):
        out = non_max_suppression(pred, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        out = non_max_suppression(pred, **kw)
    torch.cuda.synchronize()
    dt_ms = (time.perf_counter() - t0) / n * 1e3
    ncand = int(((pred[..., 4:5] * pred[..., 5:]).float() > kw['conf_thres']).sum()) if kw.get('multi_label') else int((pred[..., 4] > kw['conf_thres']).sum())
    print(f'{name:36s} {dt_ms:7.3f} ms per call  ({ncand} candidates above the threshold, {sum(o.shape[0] for o in out)} kept)')
