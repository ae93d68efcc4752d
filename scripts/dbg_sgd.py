"""debug: stock torch SGD+GradScaler vs FusedSGD+GradScaler per-step"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.cuda import amp
from multiyolov5_amd import synth
from multiyolov5_amd.models.yolo import Model
from multiyolov5_amd.utils.loss import ComputeLoss, SegmentationLosses
from multiyolov5_amd.utils.optim import FusedSGD, GradScaler
from oracle import loss_ref
from tests.util import CFG, TAGS
from tests.test_gpu_dropin import _groups
DEV = 'cuda:0'
H, W, B = 128, 256, 2
for kind in ('stock', 'fused'):
    torch.manual_seed(0)
    m = Model(os.path.join(CFG, TAGS['s_psp'])).to(DEV)
    synth.randomize_(m, seed=0)
    m.train()
    m.nc, m.gr, m.hyp = 10, 1.0, loss_ref.scaled_hyp(W, 10, 3)
    pg0, pg1, pg2 = _groups(m)
    groups = [{'params': pg0}, {'params': pg1, 'weight_decay': 5e-4}, {'params': pg2}]
    if kind == 'stock':
        opt, scaler = torch.optim.SGD(groups, lr=0.01, momentum=0.937, nesterov=True), amp.GradScaler(init_scale=256.0)
    else:
        opt, scaler = FusedSGD(groups, lr=0.01, momentum=0.937, nesterov=True), GradScaler(init_scale=256.0)
    x = synth.images(B, H, W, seed=1).to(DEV)
    t = synth.det_targets(B, 8, 10, seed=1).to(DEV)
    mk = synth.seg_targets(B, H, W, 19, seed=1).to(DEV)
    p0 = m.model[1].conv.weight
    for it in range(3):
        w_before = p0.detach().clone()
        with amp.autocast(enabled=True):
            det, seg = m(x)
            loss, _ = ComputeLoss(m)(det, t)
            sl = SegmentationLosses()(seg, mk) * B
        scaler.scale(loss * 0.6 + sl * 0.35).backward()
        gn = sum(float(p.grad.double().pow(2).sum()) for p in m.parameters()) ** 0.5
        nonfinite = sum(int((~torch.isfinite(p.grad)).sum()) for p in m.parameters())
        scaler.step(opt); scaler.update(); opt.zero_grad()
        print(kind, it, 'loss', float(loss), float(sl), 'gradnorm', gn, 'nonfinite', nonfinite, 'scale', scaler.get_scale(),
              'dw', float((p0.detach() - w_before).norm()))
