"""micro-benchmark of the multi-tensor optimizer launches on the yolov5s + PSP parameter list: FusedSGD.step (+ GradScaler check) and
ema_update at several chunk sizes; prints us per launch pair and the GB/s of the SGD pass (20 bytes per parameter)."""
import os, sys, time
import torch
sys.path.insert(0, '.')
from multiyolov5_amd.models.yolo import Model
from multiyolov5_amd.utils import optim as O
from tests.util import CFG, TAGS

m = Model(os.path.join(CFG, TAGS['s_psp'])).cuda()
ps = [p for p in m.parameters()]
n = sum(p.numel() for p in ps)
for p in ps:
    p.grad = torch.randn_like(p) * 1e-3
print('params', n, 'tensors', len(ps))
for ch in (4096, 8192, 16384, 32768, 65536):
    O.CHUNK = ch
    opt = O.FusedSGD(ps, lr=0.01, momentum=0.9, nesterov=True, weight_decay=5e-4)
    sc = O.GradScaler()
    sc._lazy(ps[0].device)
    for _ in range(3):
        sc.step(opt)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            sc.step(opt)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f'chunk {ch:6d}: check+sgd {us:7.1f} us   ({n * 24 / us / 1e6:.2f} TB/s over 24 B/param)  nchunks {opt._tab.nchunks}')
