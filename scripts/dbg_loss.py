import sys, types
import numpy as np, torch
sys.path.insert(0, '.')
from oracle import loss_ref, synth
from tests.util import golden
from multiyolov5_amd.utils.loss import ComputeLoss
DEV='cuda:0'
g = golden('losses')
for nt in (400, 800, 1500):
    rs = np.random.RandomState(11 + nt)
    B, nc = 4, 10
    shapes = ((32, 64), (16, 32), (8, 16))
    pc = [torch.from_numpy(rs.normal(0, 1.5, (B, 3, ny, nx, 5 + nc)).astype(np.float32)) for ny, nx in shapes]
    t = synth.synth_det_targets(B, max(nt // B, 1), nc, seed=9)[:nt]
    hyp = loss_ref.scaled_hyp(1024, nc, 3, label_smoothing=0.05)
    torch.set_num_threads(1)
    rl1, ri1 = loss_ref.compute_loss(pc, t, torch.from_numpy(g['anchors']), hyp)
    torch.set_num_threads(8)
    rl8, ri8 = loss_ref.compute_loss(pc, t, torch.from_numpy(g['anchors']), hyp)
    det = types.SimpleNamespace(na=3, nc=nc, nl=3, anchors=torch.from_numpy(g['anchors']).to(DEV))
    cl = ComputeLoss(types.SimpleNamespace(hyp=hyp, gr=1.0, model=[det]))
    loss, items = cl([q.to(DEV) for q in pc], t.to(DEV))
    print(nt, 'oracle1', ri1.tolist(), 'oracle8', ri8.tolist(), 'gpu', items.tolist())
    tg = loss_ref.build_targets(pc, t, torch.from_numpy(g['anchors']))
    for i,(b,a,gj,gi,tb,an,tc) in enumerate(tg):
        key = ((b*3+a)*shapes[i][0]+gj)*shapes[i][1]+gi
        print('  level', i, 'rows', len(b), 'unique cells', len(torch.unique(key)))
