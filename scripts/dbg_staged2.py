import os, sys, torch
sys.path.insert(0, '.')
from multiyolov5_amd import runtime as R
from multiyolov5_amd.models.yolo import Model
from multiyolov5_amd.utils.loss import ComputeLoss, SegmentationLosses
from oracle import loss_ref, synth
from tests.util import CFG, TAGS, synth_sd
DEV = 'cuda:0'
R.STAGED_BWD, R.FLAT_ACCUMULATE = 'force', True
torch.manual_seed(0)
m = Model(os.path.join(CFG, TAGS['s_psp']))
m.load_state_dict(synth_sd('s_psp'), strict=True)
m = m.to(DEV).train()
m.hyp, m.gr, m.nc = loss_ref.scaled_hyp(imgsz=128, nc=10, nl=3), 1.0, 10
x = synth.synth_images(2, 64, 128, seed=1).to(DEV)
targets = synth.synth_det_targets(2, 8, 10, seed=1).to(DEV)
mask = synth.synth_seg_targets(2, 64, 128, 19, seed=1).to(DEV)
cl, sl = ComputeLoss(m), SegmentationLosses()
orig = R.PlanStageFn.backward
names = ['model.0.conv.conv.weight', 'model.8.cv1.conv.weight', 'model.24.out.3.weight', 'model.25.m.0.weight']
P = dict(m.named_parameters())
def show(tag):
    torch.cuda.synchronize()
    h = list(m.__dict__['_plans'].values())[0]
    print(tag, 'acc', getattr(h, '_bwd_accumulate', None), {n.split('.')[1]: ('%.4e' % float(P[n].grad.norm()) if P[n].grad is not None else None) for n in names},
          'flat_grad norm %.4e accum norm %.4e' % (float(h.plan.flat_grad.norm()), float(h._accum_buf.norm())), 'same buf', P[names[0]].grad.untyped_storage().data_ptr() == h._accum_buf.untyped_storage().data_ptr())
for rep in range(2):
    for p in m.parameters():
        p.grad = None
    det, seg = m(x); cl(det, targets)[0].backward(); show(f'rep{rep} after det')
    det, seg = m(x); (sl(seg, mask) * 2).backward(); show(f'rep{rep} after seg')
h = list(m.__dict__['_plans'].values())[0]
print('stages', [(s['ops'], s['slices'], len(s['params'])) for s in h._stages[1]], 'n ops', h._stage_prog.n)
